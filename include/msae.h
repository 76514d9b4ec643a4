/*
 * msae.h -- C ABI of libmsae_hip.so: the MI355X (gfx950) native SAE encode / TopK / decode /
 * cache-sparsify path.  Plain pointers and sizes only; no torch types.
 *
 * Every pointer is a DEVICE pointer unless stated otherwise.  Every call is asynchronous on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream), never synchronises the
 * device, and allocates nothing: workspaces are sized by the *_ws_bytes() helpers and owned by the
 * caller (the reference calls this path inside an HF forward hook on the model's stream,
 * features/cache.py:187-204, so ops must be stream-ordered).
 *
 * Return value: 0 on success, a positive hipError_t from the HIP runtime, or a negative
 * MSAE_E* code for argument errors (the reference raises Python AssertionError on the same
 * conditions: sae/kernels.py:28-36,194-197,303-309; the Python host layer turns any non-zero
 * code into RuntimeError).  msae_error_string() describes a code.
 *
 * Reference interface each entry point replaces (paths relative to /root/reference):
 *   msae_pre_acts_f32        Sae.pre_acts                  sae_auto_interp/sae/sae.py:172-177
 *   msae_topk_f32            Sae.select_topk / torch.topk  sae_auto_interp/sae/sae.py:179-181,
 *                                                          features/cache.py:210-212
 *   msae_encode_topk         Sae.encode (fused)            sae_auto_interp/sae/sae.py:183-185
 *                            + hook edits                  features/steering.py:113-114,
 *                                                          features/patching/utils.py:43-48
 *   msae_decode_f32          Sae.decode / decoder_impl     sae_auto_interp/sae/sae.py:187-191,
 *                            TritonDecoder.forward         sae/utils.py:115-129, sae/kernels.py:178-284
 *   msae_decode_bwd_acts_f32 TritonDecoder.backward (acts) sae/kernels.py:421-425,287-400
 *   msae_decode_bwd_wdec_f32 TritonDecoder.backward (W)    sae/kernels.py:417-419,10-175
 *   msae_sparsify_*          scatter_ + Cache.add/get_nonzeros  features/cache.py:214-217,42-92
 *
 * Numerics contract: all dot products are ascending-k f32 fused
 * multiply-add chains (v_mfma_f32_32x32x2_f32 / v_fma_f32), bit-identical to oracle/sae_oracle.c.
 * msae_encode_topk selects candidates with an int8 (or bf16) MFMA pass and re-scores them with the exact f32
 * chain, so its outputs are bit-identical to msae_pre_acts_f32 + msae_topk_f32 for every token it
 * VERIFIES: all features whose coarse value plus the error band of the pair (token, feature) reaches the exact
 * k-th value were re-scored.  Tokens it cannot verify are reported in `status` and recomputed by the exact path
 * inside the same call.  WHAT THE BAND GUARANTEES depends on the mode (msae_options):
 *   default    int8 operands rounded STOCHASTICALLY with seeds drawn by the library (`dither`): for EVERY input a
 *              member of the true top-k is missed with probability <= k exp(-z^2 / 2) over the library's own randomness
 *              (a Chernoff bound of a sum of independent bounded residuals; 7e-10 per token at z = 7, k = 32; guard_z = 8:
 *              4e-13).  No assumption about the data.  The probability is over the seed of the PREPARE / refresh for the
 *              weights' residuals and, for every batch that runs an MFMA candidate pass (more than 16 tokens), for the
 *              activations' as well (the subtractive dither below: both operands are rounded against per-dim vectors fixed
 *              when the operands are prepared); the weight-stream path of <= 16 tokens rounds the activations with a seed
 *              drawn per call.  A long-running job that wants fresh randomness
 *              re-prepares (msae_encoder_refresh: one sweep over W_enc).
 *   certified  two int8 planes per operand, three MFMA segments, a DETERMINISTIC Cauchy-Schwarz band: no probability
 *              left; ~2.5x the default's step time on large batches (`certified`).
 *   exact      every token through the f32 MFMA path (`exact`); ~20x.
 *   dither off / bf16 pass: the statistical contract of ABI <= 3 (z = 7 of the rounding-noise MODEL, < 3e-13 per token under
 *              it; the model is checked on every re-scored pair) -- kept for A/B runs and for callers pinned to ABI 3.
 * Non-finite activations (+-inf / NaN, e.g. an overflowed bf16 residual stream): in every mode a token that holds one is
 * recomputed by the exact path inside the call (status 1) -- its outputs are msae_pre_acts_f32 + msae_topk_f32's -- and the
 * other tokens of the batch are unaffected (tests/test_gpu_hostile.py::test_non_finite_activations_stay_with_their_token).
 */
#ifndef MSAE_H_
#define MSAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSAE_ABI_VERSION 4

/* element type of the activation tensor x handed over by the LLM hook (sae.py:174 up-casts) */
enum { MSAE_F32 = 0, MSAE_BF16 = 1, MSAE_F16 = 2 };

/* argument errors */
enum {
  MSAE_EINVAL = -1,   /* bad shape / k / dtype code */
  MSAE_EALIGN = -2,   /* pointer or leading dimension not aligned as required */
  MSAE_EWS = -3,      /* workspace too small */
  MSAE_ENOTIMPL = -4  /* shape outside what this build supports */
};

/* ---- per-call options of the fused encoder ---------------------------------------------------------
 * The library keeps NO mutable process state: what used to be msae_set_coarse_mode / msae_set_guard_z /
 * msae_set_status_detail / msae_profile_begin (ABI 1) travels with the call.  Zero-initialise, set `size`,
 * fill what you need; NULL wherever a `const msae_options *` is taken means "all defaults".  Calls with
 * different options may run concurrently from different threads / streams.
 *   coarse_mode   operand type of the candidate pass: MSAE_COARSE_INT8 (per-token / per-feature scales, massive-
 *                 activation dims in a separately scaled k-tile), MSAE_COARSE_BF16, MSAE_COARSE_FP8 (ABI 4; OCP e4m3
 *                 operands with per-token / per-feature scales through v_mfma_f32_32x32x16_fp8_fp8, f32 accumulate --
 *                 the "fp8 MFMA encoder path" of BASELINE configs[4]; batches of any size run the 256-row tile kernel;
 *                 its prepared operands take the int8 operands' place in the buffer, so msae_encoder_prepare_opts /
 *                 _refresh with the SAME mode must precede the encode -- an encode in a mode whose operands the buffer
 *                 does not hold computes every token by the exact path; statistical contract like the bf16 pass, and
 *                 ~2x the rows to re-score: slower than int8) or MSAE_COARSE_DEFAULT
 *                 (environment MSAE_COARSE=bf16|int8|fp8, else int8).  Either way the candidates are re-scored with
 *                 the exact f32 chain, so outputs do not depend on it; the workspace size does (pass the same
 *                 options to the *_ws_bytes helper).
 *   guard_z       width z of the candidate pass's error band in standard deviations of its per-(token, feature)
 *                 rounding noise; 0 = default (environment MSAE_GUARD_Z, else 7); 0.25 <= z <= 64.  A larger z
 *                 re-scores more rows per token; results of verified tokens do not depend on it.
 *   status_detail diagnostics (tools/soak_fused.py): a token recomputed inside the call reports
 *                 status = 1 | reason << 8 (reason bits below) instead of 1, so `status & 0xFF` is the code.
 *   profile       a handle from msae_profile_create (stage timing, below) or NULL.
 *   rows_rescored (ABI 3) optional DEVICE int32[T]: per token, rounds (6 bits) << 24 | first-round rows << 12 | rows of W_enc
 *                 the exact re-score read for it (0 for tokens the large-batch re-score did not verify / paths without it) --
 *                 the measurement behind bench.py's rows_rescored_per_token; costs one store per token.  Bit 30: the first
 *                 round ran FEATURE-major (batches in which a feature is a candidate of >= ~4 tokens, e.g. k = 256 at 8192
 *                 tokens: the pairs are counting-sorted by feature, every row of W_enc comes from HBM once and the
 *                 activation rows out of the Infinity Cache -- same chains, same bits).
 *   exact         (ABI 3) != 0: msae_encode_topk[_i64] computes EVERY token by the exact path (msae_pre_acts_f32 +
 *                 msae_topk_f32 semantics through the in-call fallback: <= 1 GiB of dense scratch whatever T; status 1
 *                 for every token).  The switch for callers that cannot accept the fused path's statistical contract:
 *                 "bit-identical for every verified token, a member of the true top-k missed with probability < 3e-13
 *                 per token UNDER THE NOISE MODEL" -- the model assumes the rounding residuals of one operand are not
 *                 aligned with the other operand.  Encoder rows constructed from a token's own int8 rounding residual
 *                 (W_n ~ sign(x/sx - rint(x/sx))) violate it by construction: such a row's coarse value is low by
 *                 ~0.25 sqrt(12 d) = 55 sigma at d = 4096, it is never re-scored, no check sees it, and the int8 pass
 *                 returns status 0 with that feature missing (tests/test_gpu_hostile.py::test_row_aligned_with_a_
 *                 token_s_rounding_residual pins exactly this; the bf16 pass, whose residuals are relative roundings of
 *                 other bits, and exact = 1 return the right answer).  Trained weights cannot know a future token's
 *                 residual; weights under an adversary's control can.  Slower by ~20x on large batches.
 *                 (That paragraph describes dither = MSAE_DITHER_OFF, the behaviour of ABI <= 3.  With the dither -- the
 *                 default since ABI 4 -- the same row is found: see `dither`.)
 *   dither        (ABI 4; the word ABI 3 called `reserved`) MSAE_DITHER_DEFAULT (0: environment MSAE_DITHER=0|1, else ON),
 *                 MSAE_DITHER_ON, MSAE_DITHER_OFF.  ON: the int8 operands are rounded STOCHASTICALLY -- q = floor(v / step + r),
 *                 r uniform in [0, 1) from a counter hash of (seed, token or row, dim) -- instead of to nearest: the
 *                 activations with a fresh seed in every encode call, the weights with a fresh seed in every prepare /
 *                 refresh.  The rounding residuals are then the LIBRARY's randomness, not a property of the data: for EVERY
 *                 input (chosen without knowledge of the seeds) the error of a coarse value is a sum of independent,
 *                 zero-mean terms bounded by one step each, so by Hoeffding's inequality it exceeds z sigma' with
 *                 probability <= exp(-z^2 / 2), sigma'^2 = sw_n^2 |a_t|^2 / 4 + sx_t^2 (|W_n[in]|^2 + m_t^2 |W_n[out]|^2) / 4
 *                 (the variance PROXY of a bounded term, 3x the variance of round-to-nearest on fine data; the band the
 *                 kernels use).  A member of the true top-k is therefore missed with probability <= k exp(-z^2 / 2) per token
 *                 -- 7.3e-10 at z = 7, k = 32; 4e-13 at guard_z = 8 -- whatever the weights and activations are, including
 *                 rows built from a token's round-to-nearest residual (the pinned test now asserts the RIGHT answer).
 *                 Outputs of verified tokens do not depend on the seeds (they are the exact path's bits); which tokens fall
 *                 back may.  Costs ~sqrt(3) of band width: measured rows re-scored per token and step time in DESIGN.md
 *                 section 5.  Applies to the int8 pass (all batch sizes); the bf16 pass keeps its statistical model.
 *                 The proxy includes the cross term of the two roundings: the weights' residuals multiply the DEQUANTISED
 *                 activation, so |a_t| above reads |a_t| + sx_t sqrt(d) (ABI 4 builds before round 6 left it out).
 *                 SUBTRACTIVE DITHER (round 6; batches of more than 16 tokens: the MFMA candidate passes of csrc/gemm_mfma.h and
 *                 csrc/gemm_skinny.h; the <= 16-token weight stream keeps the band above).  The
 *                 sqrt(3) is the price of a residual whose variance f (1 - f) depends on the input.  When the dither is
 *                 subtracted again -- the operand element is taken as q - (r - 1/2) -- the residual is EXACTLY uniform on
 *                 (-1/2, 1/2] step for every input, and a uniform variable is sub-Gaussian with its own variance 1/12 as proxy:
 *                 the same bound k exp(-z^2 / 2) holds with sigma^2 = sw_n^2 (|a_t| + sx_t sqrt(d) / 2)^2 / 12 + sx_t^2 |W_n|^2 / 12,
 *                 the round-to-nearest band, now a theorem.  Subtracting is affordable because the dither is SHARED: one
 *                 vector r_x(c) for all tokens, one r_w(c) for all features (independence is needed across the dims of ONE
 *                 pair only), both derived from the seed of the prepare / refresh.  The corrections are a per-feature constant
 *                 D_n = sum_c (r_x(c) - 1/2) Wq[n][c] stored with the operands and a per-token integer E_t out of the
 *                 activation quantiser; the MFMA pass applies both at no cost (E in the multiply-add that scales the outlier
 *                 tile, D inside the epilogue's fma nesting).  Massive-activation dims keep an exact integer quotient in the
 *                 outlier tile and their remainder in their own column, so they carry the same one-step residual as every
 *                 other dim.  Measured: 45 rows re-scored per token instead of 58, re-score 0.88 -> 0.75 ms (DESIGN.md
 *                 section 5); tests/test_gpu_band.py checks the coarse value, the band and the residuals' variance pair by
 *                 pair against a numpy restatement.  A buffer prepared with dither OFF and encoded with dither ON has no
 *                 D_n: such a call computes its tokens by the exact path (status 1) -- prepare and encode with the same mode.
 *                 Environment MSAE_NO_SUBTRACT=1 keeps the non-subtractive band (A/B runs).
 *   certified     (ABI 4) != 0: msae_encode_topk[_i64] runs the CERTIFIED candidate pass: both operands as two int8 planes
 *                 (15 bits; x_lo.W_hi + x_hi.W_lo, a rounded shift by 7, then x_hi.W_hi in one int32 accumulator -- three
 *                 times the MFMA work of the default pass) and a DETERMINISTIC error band: Cauchy-Schwarz bounds of every
 *                 dropped term (the planes' rounding residuals, the lo.lo product, the shift's rounding, the f32 chain's own
 *                 gamma_d, the f32 evaluation of the bound), evaluated per (token, feature) from exact norms
 *                 (csrc/encode_cert.h derives it).  No probability and no assumption about the data is left: for EVERY input a
 *                 token reported status 0 carries exactly the exact path's top-k.  ~3x the default's step time on the bench
 *                 batch, ~9x faster than `exact`.  Needs `certified_operands` = the buffer of
 *                 msae_encoder_prepare_certified (2 d bytes per feature); `prepared` may be NULL.  Massive-activation dims have
 *                 no separate tile here: a token whose largest dim dwarfs the rest by > ~100x gets a wide band and ends in the
 *                 in-call exact path (time, never a wrong answer).  Shapes without the pass (d % 128, N % 8192) run the exact path.
 *   dither_seed   (ABI 4) 0: the library draws a seed per call (process-random base + atomic counter through a 64-bit
 *                 mixer -- the one piece of process state the library keeps); != 0: the seed of THIS call (reproducible
 *                 candidate sets: tests, A/B runs).  In a prepare / refresh it seeds the weights' rounding and the shared
 *                 dither vectors of the MFMA candidate passes; in an encode, the activations' rounding of the <= 16-token path
 *                 (and of every batch under MSAE_NO_SUBTRACT=1). */
enum { MSAE_COARSE_DEFAULT = -1, MSAE_COARSE_BF16 = 0, MSAE_COARSE_INT8 = 1, MSAE_COARSE_FP8 = 2 };
enum { MSAE_DITHER_DEFAULT = 0, MSAE_DITHER_ON = 1, MSAE_DITHER_OFF = 2 };
typedef struct msae_options {
  uint32_t size;          /* sizeof(msae_options) of the caller's header (ABI 2's 24-byte and ABI 3's 40-byte structs are
                             accepted: exact = 0 / dither_seed = 0) */
  int32_t coarse_mode;    /* MSAE_COARSE_* */
  float guard_z;          /* 0 = default */
  int32_t status_detail;  /* 0 / 1 */
  void *profile;          /* msae_profile_create handle or NULL */
  int32_t exact;          /* 0 / 1 */
  int32_t dither;         /* MSAE_DITHER_* (ABI 3: reserved, 0) */
  int32_t *rows_rescored; /* device int32[T] or NULL */
  uint64_t dither_seed;   /* 0 = drawn by the library */
  int32_t certified;      /* 0 / 1 (ABI 4) */
  int32_t reserved2;      /* 0 */
  const void *certified_operands;   /* device buffer of msae_encoder_prepare_certified, or NULL */
} msae_options;
/* Fills *opts with the defaults (coarse_mode MSAE_COARSE_DEFAULT, guard_z 0, no detail, no profile). */
void msae_options_init(msae_options *opts);

int msae_abi_version(void);
const char *msae_error_string(int code);
/* Name of the device architecture the kernels were compiled for ("gfx950"). */
const char *msae_target_arch(void);

/* ---- exact f32 path ------------------------------------------------------------------ */

/* out[T][N] = relu((x[T][d] - b_dec[d]) @ W_enc[N][d]^T + b_enc[N]), f32, row-major, dense.
 * x is row-major with element type x_dtype.  b_enc / b_dec may be NULL (treated as zeros).
 * relu != 0 applies the ReLU (Sae.pre_acts always does). */
int msae_pre_acts_f32(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                      const float *b_dec, int T, int d, int N, int relu, float *out, void *stream);

/* Canonical top-k of each row of latents[T][N]: vals[T][k] descending, ties by ascending index.
 * idx is int32 (the host layer widens to int64 where the reference API returns int64). */
size_t msae_topk_ws_bytes(int T, int N, int k);
int msae_topk_f32(const float *latents, int T, int N, int k, float *vals, int32_t *idx, void *ws,
                  size_t ws_bytes, void *stream);

/* ---- fused encoder: bf16 MFMA candidate pass + exact f32 re-score ------------------------ */

/* One-time preparation of the encoder weights (bf16 copy of W_enc + a strided sample of its rows
 * used to set the per-token candidate threshold).  `prepared` must hold
 * msae_encoder_prepared_bytes(N, d) bytes and stay valid while the encoder is used. */
size_t msae_encoder_prepared_bytes(int N, int d);
int msae_encoder_prepare(const float *W_enc, int N, int d, void *prepared, void *stream);
/* The same with options (ABI 4): `dither` / `dither_seed` choose how the int8 operands are rounded (msae_options above;
 * msae_encoder_prepare = NULL options = the defaults: dithered with a drawn seed).  Two prepares with the same non-zero seed
 * build identical operands. */
int msae_encoder_prepare_opts(const float *W_enc, int N, int d, void *prepared, const msae_options *opts, void *stream);
/* Operands of the certified pass (msae_options::certified): two int8 planes of every row of W_enc (tile-major), the rows'
 * exact norms for the deterministic band, and the biases rounded up by 2^-20 |b| (b_enc may be NULL = zeros).  Once per weight
 * (and bias) load; msae_encoder_certified_bytes(N, d) bytes (0: the shape has no certified pass -- MSAE_ENOTIMPL here). */
size_t msae_encoder_certified_bytes(int N, int d);
int msae_encoder_prepare_certified(const float *W_enc, const float *b_enc, int N, int d, void *operands, void *stream);
/* Same buffer after a weight update (one training step): rebuilds only the operands that the coarse
 * mode of `opts` reads -- the other mode's operands go stale, so call msae_encoder_prepare again before
 * encoding in the other mode. */
int msae_encoder_refresh(const float *W_enc, int N, int d, void *prepared, const msae_options *opts,
                         void *stream);
/* The buffer records which operand groups hold the current weights (bf16 | int8 | fragment-major int8); a refresh clears
 * the record of everything it does not rebuild, and an encode whose candidate pass would read a stale group computes all
 * its tokens by the exact path instead (status 1; reason 128 with status_detail) -- stale operands cost time, never a
 * wrong top-k.
 * The same for ONE following encode of T_next tokens (the training loop, train/sae/sae/trainer.py:347-401: the weights change
 * before the buffer is read again): a batch of more than 256 tokens does not read the copies the small-batch kernels use,
 * and they are left stale (a later encode of <= 256 tokens falls back to the exact path until the next refresh / prepare). */
int msae_encoder_refresh_for(const float *W_enc, int N, int d, void *prepared, int T_next, const msae_options *opts,
                             void *stream);

/* Fused Sae.encode: vals/idx[T][k] = canonical top-k of relu((x - b_dec) W_enc^T + b_enc), with
 * the reference hooks' edits of the dense latents applied before TopK:
 *   set_feature >= 0 : latents[:, set_feature] = set_value       (steering.py:113-114)
 *   zero_feature >= 0: latents[:, zero_feature] = 0               (patching/utils.py:43-48)
 * status (optional, int32[T]): 0 = fast path verified; 1 = token recomputed by the exact path
 * inside the call (every flagged token is: the fallback runs ceil(T / rows) passes over a 1 GiB
 * scratch of f32[rows][N], 2048 rows at N = 131072, sized on the device -- no host round trip).
 * Values >= 2 exist only between the kernels of one call: 2 | reason bits (4 list overflow, 8
 * threshold <= 0, 16 fewer than k candidates, 32 too many rows inside the band, 64 a re-scored pair
 * was more than 6 sigma from its coarse value: the error model does not describe this token, 128 the token's
 * shape is outside the model by a deterministic test -- the dims its int8 scale rounds to zero carry more than 4
 * bands of energy).
 * Alignment: x 16 bytes for f32, 8 bytes for the 16-bit types; large batches of a 16-BYTE-aligned x additionally take the
 * feature-major route of the exact re-score (rows_rescored bit 30; same results, less time) -- torch allocations are. */
size_t msae_encode_topk_ws_bytes(int T, int d, int N, int k, const msae_options *opts);
int msae_encode_topk(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                     const float *b_dec, const void *prepared, int T, int d, int N, int k,
                     int set_feature, float set_value, int zero_feature, float *vals, int32_t *idx,
                     int32_t *status, void *ws, size_t ws_bytes, const msae_options *opts, void *stream);
/* The same call writing 64-bit indices (EncoderOutput.top_indices is int64 in the reference: Tensor.topk,
 * sae.py:179-181); saves the widening pass on the latency path (one decode step of steering). */
int msae_encode_topk_i64(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                         const float *b_dec, const void *prepared, int T, int d, int N, int k,
                         int set_feature, float set_value, int zero_feature, float *vals, int64_t *idx,
                         int32_t *status, void *ws, size_t ws_bytes, const msae_options *opts, void *stream);

/* Exact Sae.encode of a DEVICE-SIDE LIST of tokens: for i < min(*n_rows, max_rows), t = rows[i]:
 * vals[t][k] / idx[t][k] (/ status[t] = 1, optional) = the canonical top-k of token t by msae_pre_acts_f32 +
 * msae_topk_f32 -- what msae_encode_topk returns for t, bit for bit; rows of unlisted tokens are left untouched.
 * rows / n_rows are device pointers: the work is sized on the device (ceil(max_rows / capacity) passes over a
 * <= 1 GiB dense scratch are enqueued, passes without rows exit at once), so a caller can enqueue "redo whatever the
 * previous kernel flagged" without reading the count back -- the second round of the feature-sharded engine
 * (msae/parallel.py: shards whose truncated top-k_loc list may have lost a member), inside the same forward hook
 * (features/cache.py:187-204: no host synchronisation).  No reference counterpart. */
size_t msae_encode_topk_rows_ws_bytes(int max_rows, int N);
int msae_encode_topk_rows(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                          const float *b_dec, const int32_t *rows, const int32_t *n_rows, int max_rows,
                          int d, int N, int k, int set_feature, float set_value, int zero_feature,
                          float *vals, int64_t *idx, int32_t *status, void *ws, size_t ws_bytes, void *stream);

/* ---- feature-sharded group (BASELINE configs[2]: "per-shard TopK + RCCL merge over xGMI") ------
 * The encoder's feature axis is split over G ranks; every rank sees the same T tokens.  Re-scoring a local top-k
 * exactly on every rank would multiply the HBM-bound re-score work by G, so the shards exchange CANDIDATES:
 *   msae_shard_candidates    rank g, all T tokens: candidate pass over its N/G features, then per token the C best
 *                            by upper value u as one record of msae_shard_record_bytes(C) bytes:
 *                            uint64 key[C] (order key of u << 32 | 0x7FFFFFFF - GLOBAL feature id, 0 = empty),
 *                            float z_sigma[C], float tau (largest u any OTHER feature of the shard can have; +inf when
 *                            the shard cannot bound it), float 0.  ws as for msae_encode_topk(T, d, N/G, k) with the
 *                            same options.  MSAE_ENOTIMPL for shapes without the fused pass and for MSAE_COARSE_FP8 (its
 *                            band reads the f32 weights of the batch's massive-activation dims, which a shard call does
 *                            not receive): the caller runs msae_encode_topk per shard instead.
 *   (all-to-all / all-gather of the records: rank r needs the records of its tokens from every shard)
 *   msae_rescore_candidates  rank r, its T tokens (T_valid of them real): records[G][T] -> exact top-k against the
 *                            FULL W_enc / b_enc (replicated, 2 GiB of 288 GB), same verification rule as
 *                            msae_encode_topk (every feature with u >= v_k re-scored, v_k above every shard's tau,
 *                            model check), unverifiable tokens recomputed exactly in the call.  Bit-identical to the
 *                            single-GPU msae_encode_topk.  set_feature / zero_feature are global ids (both calls). */
size_t msae_shard_record_bytes(int C);
int msae_shard_candidates(const void *x, int x_dtype, const float *b_enc_shard, const float *b_dec,
                          const void *prepared_shard, int T, int d, int N_shard, int k, int row_offset, int C,
                          int set_feature, int zero_feature, void *records, void *ws, size_t ws_bytes,
                          const msae_options *opts, void *stream);
size_t msae_rescore_candidates_ws_bytes(int T, int d, int N, int k, int G, int C);
int msae_rescore_candidates(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                            const float *b_dec, int T, int T_valid, int d, int N, int k, int G, int C,
                            const void *records, int set_feature, float set_value, int zero_feature,
                            float *vals, int64_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                            const msae_options *opts, void *stream);

/* ---- k-sparse decoder -------------------------------------------------------------------- */

/* out[A][d] = sum_j acts[A][j] * W_dec[idx[A][j]][:] + b_dec  (j-ordered f32 fma chain, entries
 * with acts == 0 skipped, kernels.py:277).  Indices outside [0, N) are skipped and, when
 * `status` (int32[1], optional) is given, flagged there (kernels.py:276 device_assert). */
int msae_decode_f32(const int32_t *idx, const float *acts, const float *W_dec, const float *b_dec,
                    int A, int k, int N, int d, float *out, int32_t *status, void *stream);
/* The same with 64-bit indices: the dtype Tensor.topk returns and the reference hands to decoder_impl
 * (sae.py:179-181,187-191) -- no narrowing copy between the encoder and the decoder. */
int msae_decode_i64_f32(const int64_t *idx, const float *acts, const float *W_dec, const float *b_dec,
                        int A, int k, int N, int d, float *out, int32_t *status, void *stream);

/* g_acts[A][k] = grad_out[A][:] . W_dec[idx[A][j]][:]   (dense-dense-sparse-out matmul).  An index outside
 * [0, N) yields g_acts = 0 for that pair and, when `status` (int32[1], optional) is given, is flagged there
 * like the forward does (kernels.py:389 device_assert). */
int msae_decode_bwd_acts_f32(const int32_t *idx, const float *grad_out, const float *W_dec, int A,
                             int k, int N, int d, float *g_acts, int32_t *status, void *stream);

/* g_W_dec[n][:] = sum over (a, j) with idx[a][j] == n of acts[a][j] * grad_out[a][:], written to
 * the WHOLE dense [N][d] buffer (the layout autograd expects for Sae.W_dec.grad; rows without a
 * pair are zero; no pre-zeroing needed).  Pairs are grouped by feature with a counting sort and
 * summed in ascending pair order (segments are sorted whatever their length), so the result is bit-reproducible.
 * Pairs whose index lies outside [0, N) contribute nothing and are flagged in `status` (int32[1], optional).
 * d % 4 == 0.  row_sumsq (optional, f32[N]) receives |g_W_dec[n][:]|^2 per row while the row is in registers: the
 * gradient-norm pass of clip_grad_norm_ (train/sae/sae/trainer.py:390) without re-reading the 2 GiB gradient
 * (msae_sum_f32 adds the rows up in a fixed order).  row_act_sum (optional, f32[N]) receives the sum of acts over the
 * row's pairs in ascending pair order: with acts = the latents' gradients (the sparse encoder backward, msae/ops.py) that is
 * the encoder-bias gradient, without index_add_'s atomics. */
size_t msae_decode_bwd_wdec_ws_bytes(int A, int k, int N);
int msae_decode_bwd_wdec_f32(const int32_t *idx, const float *acts, const float *grad_out, int A,
                             int k, int N, int d, float *g_W_dec, float *row_sumsq, float *row_act_sum,
                             int32_t *status, void *ws, size_t ws_bytes, void *stream);

/* ---- feature-cache sparsify ------------------------------------------------------------------
 * From the per-token top-k (vals/idx[B*S][k], any order) produce the reference cache's COO
 * records in its order (row, pos, feature ascending):
 *     keep (|v| > thresh) and (filter_bitmap == NULL or filter_bitmap[feature] != 0)
 *     locations[n] = (row_base + b, s, feature) int64 ; activations[n] = v
 * Two calls: _count fills counts[B*S+1] (exclusive prefix sum, counts[B*S] = nnz) on the device;
 * the caller reads nnz, allocates, then _write emits the records. */
int msae_sparsify_count(const float *vals, const int32_t *idx, int B, int S, int k, float thresh,
                        const uint8_t *filter_bitmap, int N, int64_t *counts, void *stream);
int msae_sparsify_write(const float *vals, const int32_t *idx, int B, int S, int k, float thresh,
                        const uint8_t *filter_bitmap, int N, int64_t row_base,
                        const int64_t *counts, int64_t *locations, float *activations,
                        void *stream);

/* ---- merge of per-shard results (feature-sharded encode; no reference counterpart) ----
 * gathered: int32 [G][2][T][kl], the all-gather of each rank's packed block [2][T][kl]
 * (plane 0 = f32 activation bits, plane 1 = GLOBAL feature index).  Writes the canonical top-k of
 * the G*kl candidates per token; flagged[t] (optional) = 1 when kl < k and some shard's last
 * candidate ranks inside the merged top-k (that shard may hold more members than it sent). */
int msae_merge_topk(const int32_t *gathered, int T, int G, int kl, int k, float *vals, int32_t *idx,
                    int32_t *flagged, void *stream);
/* rows[0 .. n) = the t with flags[t] != 0 in ascending order, *n_rows = n (device pointers; one small kernel): the
 * redo list msae_encode_topk_rows takes, derived identically on every rank from the same gathered data. */
int msae_compact_flags(const int32_t *flags, int T, int32_t *rows, int32_t *n_rows, void *stream);
/* msae_merge_topk for the tokens with mask[t] != 0 only, in place: the others' vals / idx rows are left as they are
 * (second round: the redone tokens' full local lists, kl = k, replace round 1's merge).  idx or idx64 may be NULL. */
int msae_merge_topk_masked(const int32_t *gathered, int T, int G, int kl, int k, const int32_t *mask,
                           float *vals, int32_t *idx, int64_t *idx64, void *stream);

/* ---- parameter-sized passes of one optimisation step (training) ----
 * msae_unit_norm_rows_f32: W[r][:] /= ||W[r][:]||_2 + eps, in place
 *                          (Sae.set_decoder_norm_to_unit_norm, sae_auto_interp/sae/sae.py:249-255).
 * msae_grad_sumsq_f32:     *accum += sum(g^2) (device scalar; the total-norm half of
 *                          clip_grad_norm_, train/sae/sae/trainer.py:390).
 * msae_adam_rows_f32:      one pass over a [rows][d] parameter: g = G * min(1, max_norm /
 *                          (sqrt(*total_sumsq) + 1e-6)) (total_sumsq NULL: no clipping); if `project`,
 *                          g -= <g, W[r]> W[r] per row (Sae.remove_gradient_parallel_to_decoder_
 *                          directions, sae.py:257-271); then Adam (torch.optim.Adam arithmetic, no
 *                          weight decay, trainer.py:139-146,395) on W, M, V in place, `step` >= 1.
 *                          G is read only. */
int msae_unit_norm_rows_f32(float *W, int N, int d, float eps, void *stream);
/* out[d] = scale * sum_n s[n] * W[n][:] in a fixed order (rows with s[n] == 0 are not read): the b_dec gradient of the sparse
 * encoder backward, -(s^T W_enc) with s = row_act_sum, as ONE streaming read of W_enc (the reference back-propagates a dense
 * [T, N] gradient through nn.Linear, sae.py:172-177; a k-row gather per token is the sparse alternative).  d % 4 == 0. */
size_t msae_weighted_row_sum_ws_bytes(int N, int d);
int msae_weighted_row_sum_f32(const float *W, const float *s, int N, int d, float scale, float *out, void *ws,
                              size_t ws_bytes, void *stream);
/* *accum += v[0] + ... + v[n - 1], summed in a fixed order (bit-reproducible): the total of row_sumsq. */
int msae_sum_f32(const float *v, size_t n, float *accum, void *stream);
/* msae_adam_rows_f32 with the NEXT step's passes over the same matrix folded into the one sweep:
 *   renorm_eps >= 0  rows of W are divided by their norm + renorm_eps after the update -- the trainer's
 *                    set_decoder_norm_to_unit_norm at the top of the next step (sae.py:249-255, trainer.py:352), bit for
 *                    bit what msae_unit_norm_rows_f32 would produce from the updated matrix; < 0: off
 *   prepared != NULL W is the ENCODER weight [N = rows][d]: its coarse-pass operands for the next encode of T_next
 *                    tokens (T_next <= 0: any batch size) are rebuilt from the updated rows, bit for bit what
 *                    msae_encoder_refresh_for / msae_encoder_refresh would build (opts: the coarse mode)
 * Both save a read (+ write) of the 2 GiB matrix per step.  Shapes the fused kernel does not take (d % 4 != 0, d > 8192)
 * run the separate passes. */
int msae_adam_rows_fused_f32(float *W, const float *G, float *M, float *V, int rows, int d,
                             const float *total_sumsq, float max_norm, int project, float lr, float beta1,
                             float beta2, float eps, int step, float renorm_eps, void *prepared, int T_next,
                             const msae_options *opts, void *stream);
int msae_grad_sumsq_f32(const float *g, size_t n, float *accum, void *stream);
int msae_adam_rows_f32(float *W, const float *G, float *M, float *V, int rows, int d,
                       const float *total_sumsq, float max_norm, int project, float lr, float beta1,
                       float beta2, float eps, int step, void *stream);

/* ---- stage timing of msae_encode_topk's fused path (measurement aid for bench.py) --------------
 * A profile handle passed in msae_options::profile makes every fused msae_encode_topk call made with it
 * record HIP events, on the stream it launches on, at the boundaries of its 6 stages:
 *   0 prep (zero + quantise x) 1 sample GEMM   2 threshold TopK   3 main int8/bf16 MFMA GEMM
 *   4 select + exact re-score 5 exact fallback of flagged tokens
 * for up to max_steps calls.  _read synchronises on the recorded events and returns stage_ms[n_steps][6]
 * (HOST pointers) and restarts the handle at step 0.  One handle must not be used by two threads at once. */
int msae_profile_create(int max_steps, void **handle);
int msae_profile_read(void *handle, float *stage_ms, int *n_steps);
int msae_profile_destroy(void *handle);

#ifdef __cplusplus
}
#endif
#endif /* MSAE_H_ */
