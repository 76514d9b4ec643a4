// encode_fused.hip -- fused Sae.encode: MFMA candidate pass (int8 or bf16) + exact f32 re-score + TopK.  This file is the HOST
// side: workspace plans, the launch sequences (run_fast, run_small, exact fallback) and the extern "C" entry points.  Kernels:
// encode_prep.h (operand preparation), gemm_mfma.h / gemm_skinny.h (candidate passes), encode_rescore.h (select + exact re-score,
// shard records), encode_small.h (T <= 16), topk.hip / encode_f32.hip (exact path); shared layout / options: encode_defs.h;
// build-time knobs and probes: tuning.h.
//
// Replaces Sae.encode = select_topk(pre_acts(x)) (reference sae/sae.py:172-185) without ever
// writing the dense [T][N] latents (512 KiB/token at N = 131072) to HBM.
//
// Pipeline per call (all on one stream, no host synchronisation):
//   1. prep_x        a32[T][d] = f32(x) - b_dec  (the exact SAE input)
//      int8 pass:    column max over the batch -> outlier dims -> per-token scale sx[t], int8 rows xq,
//                    outlier dims in their own 128-wide k-tile at scale m[t]*sx[t]; the matching
//                    columns of Wq are gathered into an outlier tile.  (bf16 pass: xb = bf16(a32).)
//   2. gemm<DENSE>   coarse pre-acts of a 1/32 strided SAMPLE of the features -> [T][S] f32
//   3. kth value     tau[t] = r-th largest sample value: ~32*r features of the full width exceed it; the sample
//                    features above tau start the token's candidate list (KthPush / sample_push_kernel)
//   4. gemm<THRESH>  THE DOMINANT KERNEL (gemm_mfma.h): [T][d] x [d][N] on the matrix cores -- for batches of more
//                    than 256 tokens over the 31/32 of the features the sample pass has not scored (main_row);
//                    epilogue: scales, +b_enc, compare with tau[t], append (feature, coarse) of the
//                    rare survivors to a per-token candidate list.  Roofline: MFMA, 2*d*N op/token (in practice the
//                    package power limit: DESIGN.md section 5).
//   5. select_rescore per token: order candidates by their UPPER value u, re-score the best ones
//                    with the exact ascending-k f32 fma chain over the f32 W_enc rows, take the
//                    canonical top-k, and verify that EVERY feature whose u reaches the exact
//                    k-th value v_k has been re-scored (at most one extension round).  Tokens
//                    that fail (list overflow, tau <= 0, model violation ...) are flagged.
//   6. exact path    flagged tokens (normally none) are recomputed by encode_f32 + topk through a
//                    device-side row list; their results overwrite step 5's.
//
// Verification model.  The coarse value c(t,n) of the candidate pass differs from the exact
// pre-activation p(t,n) by rounding noise whose variance is known per (token, feature) PAIR:
//   int8:  p - c = sum_c [ a_c sw_n eps_c + sx_t delta_c w_c ],  eps, delta = rounding residuals in
//          (-1/2, 1/2] steps (delta in m_t steps on the outlier dims), so
//          sigma^2(t,n) = sw_n^2 |a_t|^2 / 12 + sx_t^2 (|W_n[in]|^2 + m_t^2 |W_n[out]|^2) / 12
//   bf16:  p - c = -sum_c a_c w_c (da_c + dw_c), relative roundings of variance 2.75e-6 each, so
//          sigma^2(t,n) <= 5.5e-6 |a_t|_4^2 |W_n|_4^2   (Cauchy-Schwarz on sum a_c^2 w_c^2)
// both of the separable form  z^2 sigma^2 = P_t Q_n + R_t (Si_n + M_t So_n).  Every stage works on
// u = c + z sigma (z = 7 by default): the sample threshold tau is a rank statistic of u, the GEMM
// emits u > tau, and a token is verified when all features with u >= v_k were re-scored exactly --
// a feature is then missed only if its own error exceeds z of ITS sigma (heterogeneous rows: spiky,
// large-norm or near-dead encoder rows carry their own band).  Rows whose bulk lies below one int8
// step (max > 127 rms) are rounded stochastically (hash dither), which keeps the residual unbiased
// whatever direction the activations have, at variance sw^2/4.  The model is CHECKED on every
// re-scored pair: |p - c| > 6 sigma flags the token (reason 64) and it goes to the exact path.
// DESIGN.md section 4 gives the failure-probability arithmetic.
//
// Outputs are therefore bit-identical to msae_pre_acts_f32 + msae_topk_f32 whenever the token
// verifies, and ARE that path's outputs when it does not -- whichever operand type ran step 4.
#include <new>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_mfma.h"
#include "gemm_skinny.h"
#include "encode_defs.h"
#include "encode_prep.h"
#include "encode_rescore.h"
#include "encode_small.h"
#include "encode_cert.h"

int msae_pre_acts_launch(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                         const float *b_dec, const int *rows, const int *n_rows, int T, int d, int N,
                         int relu, float *out, int ld_out, hipStream_t s);
int msae_topk_launch(const float *latents, int T, int N, int k, int ld, const int *n_rows,
                     float *vals, int32_t *idx, hipStream_t s, const TopkExtra &ex = TopkExtra());
bool msae_kth_value_launch(const float *rows, int T, int S, int ld, int r, float *out, int out_ld,
                           int out_col, hipStream_t s, const KthPush &push = KthPush());

namespace {

// ---- MFMA GEMM: gemm_mfma.h.  Tile choice from tools/gemm_sweep on MI355X (T=8192, d=4096,
// N=131072): 256x256 tiles of 128-B k-rows, 2-slot ring, 8 waves as 2x4.
using GemmBf16 = GemmCfg<256, 256, 2, 2, 4, false>;
using GemmI8 = GemmCfg<256, 256, 2, 2, 4, true>;
using GemmI8Cert = GemmCfg<256, 256, 2, 2, 4, true, 32>;   // msae_options::certified (encode_cert.h)
using GemmF8 = GemmCfg<256, 256, 2, 2, 4, false, 64>;      // MSAE_COARSE_FP8: e4m3 operands (BASELINE configs[4])
constexpr int G_BM = GemmBf16::BM;

// three scratch ranges in one launch (candidate counters, flag list, column maxima)
__global__ void zero3_i32_kernel(int *p0, size_t n0, int *p1, size_t n1, int *p2, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n0; i += (size_t)gridDim.x * 256) p0[i] = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) p1[i] = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) p2[i] = 0;
}

__global__ void zero_i32_kernel(int *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

// hook edits on dense rows (exact path): latents[:, set_feature] = set_value; [:, zero_feature] = 0
__global__ void edit_dense_kernel(float *dense, int ld, int rows, const int *n_rows, int set_feature,
                                  float set_value, int zero_feature) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int R = n_rows ? min(rows, *n_rows) : rows;
  if (r >= R) return;
  if (set_feature >= 0) dense[(size_t)r * ld + set_feature] = set_value;
  if (zero_feature >= 0) dense[(size_t)r * ld + zero_feature] = 0.f;
}

// list[0 .. T) = 0 .. T - 1, list[T] = T (the count), the words behind it 0: "every token is flagged" (msae_options::exact)
__global__ void iota_list_kernel(int *list, int T, int n_total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_total) list[i] = i < T ? i : (i == T ? T : 0);
}

// counts[c] = number of flagged tokens in pass c of the exact fallback
__global__ void fallback_counts_kernel(const int *n_flagged, int fb_cap, int chunks, int *counts) {
  const int nf = *n_flagged;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    const int left = nf - c * fb_cap;
    counts[c] = left < 0 ? 0 : (left > fb_cap ? fb_cap : left);
  }
}

// msae_shard_candidates: where this shard's records go
struct ShardOut { unsigned char *recs; int C; int row_offset; };

// ---- stage profiling (bench.py roofline): HIP events recorded on the launch stream ------------------
constexpr int PROF_MARKS = 7;  // boundaries of: prep | sample gemm | tau topk | main gemm | rescore | fallback
struct ProfState {
  unsigned magic = 0x50524F46u;   // "PROF"
  int max_steps = 0, step = 0;
  hipEvent_t *ev = nullptr;
};

inline void prof_mark(ProfState *pf, int i, hipStream_t s) {
  if (pf && pf->step < pf->max_steps) (void)hipEventRecord(pf->ev[pf->step * PROF_MARKS + i], s);
}
inline void prof_step(ProfState *pf) {
  if (pf && pf->step < pf->max_steps) ++pf->step;
}

// ---- workspace carving -------------------------------------------------------------------------
struct FusedPlan {
  bool fast, i8, small, fm, f8;
  size_t off_fmcount, off_fmtarget, off_fmkeys, off_fmpairs, off_fmpre, off_fmdefer;
  size_t off_xhi, off_xlo, off_skeys, off_sviol, off_surv, off_sbound, off_scand, off_stau;
  int Tp, S, r, cap, r_max, fb_cap, fb_chunks;
  size_t off_rowe, off_cds, off_cds_s, off_cds_p;   // subtractive dither: (E, m) per token, Ds per column in the three column orders
  size_t off_xq, off_xqo, off_rowc, off_refs, off_colc, off_colc_s, off_colc_p, off_colmax, off_odims, off_isout, off_wqo, off_wqos;
  size_t off_xb, off_a32, off_sample, off_tauv, off_taui, off_cnt, off_cand, off_segcnt, off_segcand, off_flag, off_fbdense, off_dense, bytes;
  int segs;   // > 1: the candidate passes append to segmented lists (compact_candidates_kernel joins them)
};

inline FusedPlan make_plan(int T, int d, int N, int k, int mode, int shard_C = 0, bool cert = false) {
  FusedPlan p{};
  p.fast = fast_shape_ok(N, d) && T > EXACT_T_MAX && k <= 256 && k >= 1;
  if (cert && !cert_shape_ok(N, d)) p.fast = false;      // shapes without the certified pass: the exact path (certainly certified)
  if (cert) mode = 1;
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += msae_align_up(b, 256); return at; };
  if (p.fast) {
    p.Tp = (T + G_BM - 1) / G_BM * G_BM;
    p.S = N / SAMPLE_STRIDE;
    // tau = r-th largest of the 1/32 sample: ~32*r survivors, Gamma(r)-distributed.  r = 16 keeps
    // P(fewer than ~2k survivors) and P(overflow) below 1e-9 per token (r = 8 flagged 3 of 8192)
    p.r = k / 8 > 16 ? k / 8 : 16;         // k = 256: r = 32 -> ~1024 survivors, capacity 4096
    if (k < 32) p.r = k / 2 > 8 ? k / 2 : 8;   // small k (a shard's k_loc): ~k + band rows are needed, 256+ survive
    // A shard of a feature-sharded group only has to deliver its C best: 256+ survivors are plenty (its share of
    // the needed set is below C / 1.5 by construction of C, P(Gamma(8) x 32 below that) ~ 1e-5), and the default
    // would put 512 survivors per token into 1/G of the columns -- at G = 8 that is 2048 per output tile, the
    // size of the epilogue's LDS queue.
    if (shard_C > 0) { const int rs = shard_C / 8 > 8 ? shard_C / 8 : 8; if (rs < p.r) p.r = rs; }
    p.cap = next_pow2(128 * p.r);           // 4x the expected count
    p.i8 = mode == 1 && i8_shape_ok(N, d);
    p.f8 = mode == 2 && i8_shape_ok(N, d);     // (other shapes: the bf16 pass, whose operands an fp8 prepare builds as well)
    p.small = !cert && p.i8 && small_shape_ok(T, d, N, k) && getenv("MSAE_NO_SMALL_PATH") == nullptr;
    // most rows one token may read before it is handed to the exact path (the needed set is ~k + 10:
    // reaching this means the band is not separating anything); at least k + 4 (first-round minimum)
    p.r_max = k <= 64 ? 8 * k : 3 * k;
    if (p.r_max < k + 4) p.r_max = k + 4;
    if (p.r_max > p.cap) p.r_max = p.cap;
    if (p.i8 || p.f8) {   // the batch's massive-activation dims and the per-call column constants of the band
      p.off_colc = take((size_t)N * 16);
      p.off_colc_s = take((size_t)p.S * 16);
      p.off_colc_p = take(p.i8 ? (size_t)N * 16 : 0);
      p.off_colmax = take((size_t)d * 4 * COLMAX_PARTS);
      p.off_odims = take((size_t)(MAX_OUT + 1) * 4);
      p.off_isout = take((size_t)d);
    }
    if (p.i8) {
      p.off_xq = take((size_t)p.Tp * d * (cert ? 2 : 1));   // (certified: two planes per token row)
      p.off_xqo = take((size_t)p.Tp * MAX_OUT);
      p.off_wqo = take((size_t)N * MAX_OUT);
      p.off_wqos = take((size_t)p.S * MAX_OUT);
    }
    p.off_rowc = take((size_t)p.Tp * 16);
    p.off_rowe = take(p.i8 ? (size_t)p.Tp * 8 : 0);
    p.off_cds = take(p.i8 ? (size_t)N * 4 : 0);
    p.off_cds_s = take(p.i8 ? (size_t)p.S * 4 : 0);
    p.off_cds_p = take(p.i8 ? (size_t)N * 4 : 0);
    p.off_refs = take(256);
    if (p.small) {
      p.off_xhi = take((size_t)T * d);
      p.off_xlo = take((size_t)T * d);
      p.off_skeys = take((size_t)T * 128 * 8);
      p.off_surv = take((size_t)T * SMALL_SURV * 8);
      p.off_sbound = take((size_t)T * SMALL_GRID * 4);
      p.off_scand = take((size_t)T * 128 * 8);
      p.off_stau = take((size_t)T * 4);
      p.off_sviol = take((size_t)T * 2 * 4);            // model-check flags [T] | finished-wave counters [T]
    }
    p.off_xb = take(p.i8 ? 256 : (size_t)p.Tp * d * (p.f8 ? 1 : 2));
    p.off_a32 = take((size_t)T * d * 4);
    p.off_sample = take((size_t)T * p.S * 4);
    const size_t tau_n = (size_t)T * p.r;
    p.off_tauv = take(tau_n * 4);
    p.off_taui = take(tau_n * 4);
    p.off_cnt = take((size_t)T * 4);
    p.segs = (T <= 256 && p.cap >= 1024) ? 8 : 1;      // few tokens: hundreds of appends per list counter (GemmEpilogue::segs)
    p.off_segcnt = take(p.segs > 1 ? (size_t)T * p.segs * 4 : 0);   // right behind cnt: zeroed with it
    p.off_cand = take((size_t)T * p.cap * 8);
    p.off_segcand = take(p.segs > 1 ? (size_t)T * p.cap * 8 : 0);
    p.fb_cap = fallback_capacity(T, N);
    p.fb_chunks = (T + p.fb_cap - 1) / p.fb_cap;
    p.off_flag = take(((size_t)T + 64 + p.fb_chunks) * 4);   // token list [T] | count | per-pass counts
    p.off_fbdense = take((size_t)p.fb_cap * N * 4);
    p.fm = shard_C == 0 && fm_shape_ok(T, k, N, d, p.r_max);    // feature-major first round of the re-score (encode_rescore.h)
    if (p.fm) {
      p.off_fmcount = take(((size_t)N + 64 + (N + FM_SCAN_BLOCK - 1) / FM_SCAN_BLOCK) * 4);   // counts [N] | total | block sums
      p.off_fmtarget = take((size_t)T * 4);
      p.off_fmkeys = take((size_t)T * p.r_max * 8);
      p.off_fmpairs = take(((size_t)T * p.r_max + (size_t)N * 16) * 8);   // slots: the pairs + every feature's padding to whole groups
      p.off_fmpre = take((size_t)T * p.r_max * 4);
      p.off_fmdefer = take((size_t)T * 2 * 4);              // tokens the LEAN launch of PHASE 1 / 2 left to the full-size one
    }
  } else {
    p.off_dense = take((size_t)T * N * 4);
  }
  p.bytes = o;
  return p;
}


// exact recompute of a device-side list of tokens, fb_cap at a time (device-side counts; passes without work exit
// immediately): list[0 .. *n_list) of token rows, pass_counts[fb_chunks] scratch, dense f32[fb_cap][N] scratch
template <int DT>
int run_exact_rows(const void *x, const float *W_enc, const float *b_enc, const float *b_dec, const int *list,
                   const int *n_list, int *pass_counts, float *dense, int fb_cap, int fb_chunks, int d, int N, int k,
                   int set_feature, float set_value, int zero_feature, float *vals, IdxOut idx, int32_t *status,
                   int detail, hipStream_t s) {
  if (fb_chunks > 1)
    hipLaunchKernelGGL(fallback_counts_kernel, dim3(1), dim3(64), 0, s, n_list, fb_cap, fb_chunks, pass_counts);
  for (int c = 0; c < fb_chunks; ++c) {
    // one pass covers every token (T <= fb_cap): the list's count itself is the pass's row count
    const int *rows = list + (size_t)c * fb_cap, *n_rows = fb_chunks > 1 ? pass_counts + c : n_list;
    int rc = msae_pre_acts_launch(x, DT, W_enc, b_enc, b_dec, rows, n_rows, fb_cap, d, N, 1, dense, N, s);
    if (rc) return rc;
    if (set_feature >= 0 || zero_feature >= 0)
      hipLaunchKernelGGL(edit_dense_kernel, dim3((fb_cap + 255) / 256), dim3(256), 0, s, dense, N, fb_cap, n_rows,
                         set_feature, set_value, zero_feature);
    // the exact results go straight to the listed tokens' rows of the outputs (row map = the list)
    TopkExtra ex;
    ex.idx64 = idx.i64; ex.row_map = rows; ex.status = status; ex.detail = detail;
    rc = msae_topk_launch(dense, fb_cap, N, k, N, n_rows, vals, idx.i32, s, ex);
    if (rc) return rc;
  }
  return 0;
}

// ... of the tokens the fused path flagged
template <int DT>
int run_exact_fallback(const void *x, const float *W_enc, const float *b_enc, const float *b_dec, int T, int d, int N,
                       int k, int set_feature, float set_value, int zero_feature, float *vals, IdxOut idx,
                       int32_t *status, unsigned char *ws, const FusedPlan &pl, int detail, hipStream_t s) {
  int *flagged = reinterpret_cast<int *>(ws + pl.off_flag);
  return run_exact_rows<DT>(x, W_enc, b_enc, b_dec, flagged, flagged + T, flagged + T + 64,
                            reinterpret_cast<float *>(ws + pl.off_fbdense), pl.fb_cap, pl.fb_chunks, d, N, k, set_feature,
                            set_value, zero_feature, vals, idx, status, detail, s);
}

inline int dot4_max_small() {   // largest T of the dot4 weight stream (tuning knob; the MFMA stream takes the rest of the small path)
  static const int v = [] { const char *e = getenv("MSAE_SMALL_DOT4_MAX"); return e ? atoi(e) : SMALL_DOT4_PREF; }();
  return v;
}

template <int DT>
int run_small(const void *x, const float *W_enc, const float *b_enc, const float *b_dec, const Prepared &pp,
              const unsigned char *prepared, int T, int d, int N, int k, int set_feature, float set_value,
              int zero_feature, float *vals, IdxOut idx, int32_t *status, unsigned char *ws, const FusedPlan &pl,
              const CallOpts &co, hipStream_t s) {
  // zz12: z^2 x the W-side variance of one rounding inside P_t (the dither's factor 3 is in Q_n); zzx: ... of the x side
  const float z = co.z, zz12 = z * z / 12.f, zzx = z * z * x_round_var(co.seed != 0ull);
  float *a32 = reinterpret_cast<float *>(ws + pl.off_a32);
  signed char *xhi = reinterpret_cast<signed char *>(ws + pl.off_xhi);
  signed char *xlo = reinterpret_cast<signed char *>(ws + pl.off_xlo);
  f32x4 *rowc = reinterpret_cast<f32x4 *>(ws + pl.off_rowc);
  unsigned long long *surv = reinterpret_cast<unsigned long long *>(ws + pl.off_surv);
  unsigned *bound = reinterpret_cast<unsigned *>(ws + pl.off_sbound);
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + pl.off_scand);
  float *tau = reinterpret_cast<float *>(ws + pl.off_stau);
  unsigned long long *exact = reinterpret_cast<unsigned long long *>(ws + pl.off_skeys);
  int *viol = reinterpret_cast<int *>(ws + pl.off_sviol);
  int *done = viol + T;
  int *flagged = reinterpret_cast<int *>(ws + pl.off_flag);
  int *n_flagged = flagged + T;
  const signed char *wq = reinterpret_cast<const signed char *>(prepared + pp.off_wq);
  const signed char *wqf = reinterpret_cast<const signed char *>(prepared + pp.off_wqf);     // fragment-major copies (MFMA stream)
  const signed char *wqsf = reinterpret_cast<const signed char *>(prepared + pp.off_wqsf);
  const f32x4 *wstat = reinterpret_cast<const f32x4 *>(prepared + pp.off_wstat);
  prof_mark(co.prof, 0, s);
  const unsigned *valid = reinterpret_cast<const unsigned *>(prepared + offsetof(Prepared, valid));
  const unsigned need = (T > dot4_max_small() && d <= 4096) ? (PREP_I8 | PREP_FRAG) : PREP_I8;
  hipLaunchKernelGGL(prep_small_kernel<DT>, dim3(T), dim3(256), 0, s, x, b_dec, d, a32, xhi, xlo, rowc, zz12, viol, 2 * T,
                     flagged, T + 64 + pl.fb_chunks, valid, need, T, co.seed);
  prof_mark(co.prof, 1, s);
  prof_mark(co.prof, 2, s);
  prof_mark(co.prof, 3, s);
  const int skip_a = set_feature >= 0 ? set_feature : -1, skip_b = zero_feature >= 0 ? zero_feature : -1;
#define MSAE_GEMV(DSEG, TT)                                                                                        \
  hipLaunchKernelGGL((gemv_small_kernel<DSEG, TT>), dim3(SMALL_GRID), dim3(256), 0, s, wq, wstat, b_enc, N, T, xhi, xlo, \
                     rowc, zzx, skip_a, skip_b, surv, bound)
  const int dseg = d / 1024;
  int n_surv = SMALL_SURV, n_bound = SMALL_GRID;
  const int dot4_max = dot4_max_small();
  if (T > dot4_max && d <= 4096) {
    static int n_cu = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return cus > 0 ? cus : 256;
    }();
    const int grid_m = n_cu < SMALL_GRID ? n_cu : SMALL_GRID;
    n_surv = grid_m * SMALL_MF_EMIT; n_bound = grid_m;
    const size_t smem_m = (size_t)2 * 16 * (d + 16) + (size_t)16 * 8 * (SMALL_MF_EMIT + 1) * 8;
#define MSAE_GEMV_M(DSEG)                                                                                          \
  do {                                                                                                             \
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)gemv_mfma_kernel<DSEG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int)smem_m));                                                                \
    hipLaunchKernelGGL(gemv_mfma_kernel<DSEG>, dim3(grid_m), dim3(512), smem_m, s, wqf, wqsf, wstat, b_enc, N, T, xhi, xlo, rowc, \
                       zzx, skip_a, skip_b, surv, bound);                                                          \
  } while (0)
    switch (dseg) { case 1: MSAE_GEMV_M(1); break; case 2: MSAE_GEMV_M(2); break; case 4: MSAE_GEMV_M(4); break;
                    default: return MSAE_ENOTIMPL; }
#undef MSAE_GEMV_M
  } else if (T == 1) {
    switch (dseg) { case 1: MSAE_GEMV(1, 1); break; case 2: MSAE_GEMV(2, 1); break; case 4: MSAE_GEMV(4, 1); break;
                    case 8: MSAE_GEMV(8, 1); break; default: return MSAE_ENOTIMPL; }
  } else if (T == 2) {
    switch (dseg) { case 1: MSAE_GEMV(1, 2); break; case 2: MSAE_GEMV(2, 2); break; case 4: MSAE_GEMV(4, 2); break;
                    case 8: MSAE_GEMV(8, 2); break; default: return MSAE_ENOTIMPL; }
  } else {
    switch (dseg) { case 1: MSAE_GEMV(1, 4); break; case 2: MSAE_GEMV(2, 4); break; case 4: MSAE_GEMV(4, 4); break;
                    default: return MSAE_ENOTIMPL; }
  }
#undef MSAE_GEMV
  hipLaunchKernelGGL(select_small_kernel, dim3(T), dim3(1024), 0, s, surv, bound, cand, tau, n_surv, n_bound);
  prof_mark(co.prof, 4, s);
#define MSAE_RESCORE(DSEG)                                                                                         \
  hipLaunchKernelGGL(rescore_small_kernel<DSEG>, dim3(SMALL_RMAX, T), dim3(64), 0, s, a32, W_enc, b_enc, k, cand, tau, wstat, \
                     rowc, zzx, z * z, guard_z_check2(co.seed != 0ull), set_feature, set_value, exact, viol, done, vals, idx, status, flagged, n_flagged)
  switch (dseg) { case 1: MSAE_RESCORE(1); break; case 2: MSAE_RESCORE(2); break; case 4: MSAE_RESCORE(4); break;
                  case 8: MSAE_RESCORE(8); break; default: return MSAE_ENOTIMPL; }
#undef MSAE_RESCORE
  prof_mark(co.prof, 5, s);
  int rc = run_exact_fallback<DT>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx,
                                  status, ws, pl, co.detail, s);
  if (rc) return rc;
  prof_mark(co.prof, 6, s);
  prof_step(co.prof);
  return msae_launch_status();
}

// select + exact re-score of the candidate lists (RescoreArgs filled by the caller): token-major, or with the FEATURE-major first
// round where the plan has it and the cost model says it pays (encode_rescore.h).  Shared by run_fast and run_cert.
template <int DT>
int rescore_stage(RescoreArgs &ra, const FusedPlan &pl, const void *x, const float *b_dec, const float *W_enc,
                  const float *b_enc, const float *a32, unsigned long long *cand, unsigned char *ws, int T, int d, int N, int k,
                  hipStream_t s) {
  const int nrp = next_pow2(pl.r_max + 1);
  const size_t smem = ((size_t)pl.cap + nrp) * 8 + 64;
  int lrc;
  // (fm_dot_kernel reads x in 16-B pieces; the entry points ask 8 B of a 16-bit x)
  if (pl.fm && msae_aligned(x, 16) && fm_pays(T, k, N, d, DT == MSAE_F32 ? 4 : 2)) {
    int *fcount = reinterpret_cast<int *>(ws + pl.off_fmcount);
    int2 *pairs = reinterpret_cast<int2 *>(ws + pl.off_fmpairs);
    float *fpre = reinterpret_cast<float *>(ws + pl.off_fmpre);
    ra.fm_count = fcount; ra.fm_target = reinterpret_cast<int *>(ws + pl.off_fmtarget);
    ra.fm_keys = reinterpret_cast<unsigned long long *>(ws + pl.off_fmkeys); ra.fm_pre = fpre; ra.fm_rcap = pl.r_max; ra.fm_cand = cand;
    ra.fm_rank = reinterpret_cast<int *>(fpre);
    ra.fm_defer = reinterpret_cast<int *>(ws + pl.off_fmdefer);
    MSAE_HIP_TRY(hipMemsetAsync(ra.fm_defer, 0, (size_t)T * 2 * 4, s));
    MSAE_HIP_TRY(hipMemsetAsync(fcount, 0, ((size_t)N + 1) * 4, s));
    lrc = launch_select_rescore<false, 1>(ra, T, k, smem, (const float *)a32, W_enc, s);
    if (lrc) return lrc;
    const int scan_blocks = (N + FM_SCAN_BLOCK - 1) / FM_SCAN_BLOCK, G = fm_group_lanes(T, k, N);
    hipLaunchKernelGGL(fm_blocksum_kernel, dim3(scan_blocks), dim3(256), 0, s, fcount, N, G, fcount + N + 64);
    hipLaunchKernelGGL(fm_scan_kernel, dim3(scan_blocks), dim3(256), 0, s, fcount, N, G, fcount + N + 64, pairs);
    hipLaunchKernelGGL(fm_scatter_kernel, dim3(T), dim3(256), 0, s, ra.fm_target, ra.fm_keys, ra.fm_rank, pl.r_max, fcount, pairs);
    const long max_slots = (long)T * pl.r_max + (long)N * (G - 1);
    const dim3 dgrid((unsigned)((max_slots + 63) / 64));
    if (G == 16) hipLaunchKernelGGL((fm_dot_kernel<DT, 16>), dgrid, dim3(64), 0, s, x, b_dec, W_enc, b_enc, pairs, fcount + N, d, pl.r_max, fpre);
    else hipLaunchKernelGGL((fm_dot_kernel<DT, 4>), dgrid, dim3(64), 0, s, x, b_dec, W_enc, b_enc, pairs, fcount + N, d, pl.r_max, fpre);
    lrc = launch_select_rescore<false, 2>(ra, T, k, smem, (const float *)a32, W_enc, s);
  } else {
    lrc = launch_select_rescore<false>(ra, T, k, smem, (const float *)a32, W_enc, s);
  }
  return lrc;
}

template <int DT>
int run_fast(const void *x, const float *W_enc, const float *b_enc, const float *b_dec,
             const Prepared &pp, const unsigned char *prepared, int T, int d, int N, int k,
             int set_feature, float set_value, int zero_feature, float *vals, IdxOut idx,
             int32_t *status, unsigned char *ws, const FusedPlan &pl, const CallOpts &co, hipStream_t s,
             const ShardOut *shard = nullptr) {
  unsigned short *xb = reinterpret_cast<unsigned short *>(ws + pl.off_xb);
  float *a32 = reinterpret_cast<float *>(ws + pl.off_a32);
  float *sample = reinterpret_cast<float *>(ws + pl.off_sample);
  float *tauv = reinterpret_cast<float *>(ws + pl.off_tauv);
  int32_t *taui = reinterpret_cast<int32_t *>(ws + pl.off_taui);
  int *cnt = reinterpret_cast<int *>(ws + pl.off_cnt);
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + pl.off_cand);
  int *flagged = reinterpret_cast<int *>(ws + pl.off_flag);
  int *n_flagged = flagged + T;
  const unsigned short *wb = reinterpret_cast<const unsigned short *>(prepared + pp.off_wb);
  const unsigned short *wsamp = reinterpret_cast<const unsigned short *>(prepared + pp.off_ws);
  const unsigned *valid = reinterpret_cast<const unsigned *>(prepared + offsetof(Prepared, valid));
  prof_mark(co.prof, 0, s);
  // producers of the candidate lists write the segmented lists when the plan has them (compact_candidates_kernel joins them)
  int *pcnt = pl.segs > 1 ? reinterpret_cast<int *>(ws + pl.off_segcnt) : cnt;
  unsigned long long *pcand = pl.segs > 1 ? reinterpret_cast<unsigned long long *>(ws + pl.off_segcand) : cand;
  const int seg_cap = pl.cap / pl.segs;
  const size_t n_cnt = pl.segs > 1 ? (pl.off_segcnt - pl.off_cnt) / 4 + (size_t)T * pl.segs : (size_t)T;
  hipLaunchKernelGGL(zero3_i32_kernel, dim3(64), dim3(256), 0, s, cnt, n_cnt, flagged, (size_t)T + 64 + pl.fb_chunks,
                     (pl.i8 || pl.f8) ? reinterpret_cast<int *>(ws + pl.off_colmax) : (int *)nullptr,
                     (pl.i8 || pl.f8) ? (size_t)d * COLMAX_PARTS : (size_t)0);
  if (!pl.i8 && !pl.f8)
    hipLaunchKernelGGL(prep_x_kernel<DT>, dim3(2048), dim3(256), 0, s, x, b_dec, T, pl.Tp, d, xb, a32);

  GemmOperands op_main{}, op_samp{};
  // (see run_small; fp8: the band's absolute-grid terms, encode_defs.h)
  float z = co.z, zz12 = z * z / 12.f, zzx = pl.f8 ? z * z * FP8_ABS_VAR : z * z * x_round_var(pl.i8 && co.seed != 0ull);
  bool sd = false;                 // subtractive dither (encode_defs.h): the large-batch int8 pass under msae_options::dither
  f32x4 *rowc = reinterpret_cast<f32x4 *>(ws + pl.off_rowc);
  const f32x4 *colc, *colc_s;      // error-band column constants of the main / sample pass
  f32x4 *cc_perm = nullptr;        // ... of the main pass in its own column order when it leaves the sample rows out
  bool skip_sample = false;
  int skinny = 0;                  // 64 / 128: token rows of the weight-stream kernel's tile (gemm_skinny.h); 0: gemm_mfma.h
  if (pl.i8) {
    signed char *xq = reinterpret_cast<signed char *>(ws + pl.off_xq);
    signed char *xqo = reinterpret_cast<signed char *>(ws + pl.off_xqo);
    f32x4 *cc_main = reinterpret_cast<f32x4 *>(ws + pl.off_colc);
    f32x4 *cc_samp = reinterpret_cast<f32x4 *>(ws + pl.off_colc_s);
    unsigned *colmax = reinterpret_cast<unsigned *>(ws + pl.off_colmax);
    int *odims = reinterpret_cast<int *>(ws + pl.off_odims);
    unsigned char *is_out = ws + pl.off_isout;
    signed char *wqo = reinterpret_cast<signed char *>(ws + pl.off_wqo);
    signed char *wqos = reinterpret_cast<signed char *>(ws + pl.off_wqos);
    const signed char *wq = reinterpret_cast<const signed char *>(prepared + pp.off_wq);
    const signed char *wqs = reinterpret_cast<const signed char *>(prepared + pp.off_wqs);
    // tile-major operands for the candidate GEMM (MSAE_GEMM_ROWMAJOR=1: the row-major copies, for A/B runs)
    // one row of output tiles (T <= 256) streams Wq from HBM once and keeps round 2's row-major operands + unstaggered
    // issue: tile-major + stagger measured 2-3 % slower there (profiles/r03_ab_small_T.txt)
    const int tile_major = pl.Tp > G_BM ? gemm_layout() : 0;
    // up to 128 tokens: the weight-stream kernel (gemm_skinny.h) runs both candidate passes: xq row-major, Wq fragment-major
    if (T <= 256 && tile_major == 0 && gemm_layout() == 1 && d % 1024 == 0 && N % (SAMPLE_STRIDE * 256) == 0 &&
        getenv("MSAE_NO_SKINNY") == nullptr)
      skinny = T <= 64 ? 64 : (T <= 128 ? 128 : 256);
    const bool w_packed = tile_major == 1 || skinny != 0;   // the W side of the candidate passes reads the tile-major copies
    const int ychunks = T >= 32 ? (T / 16 < 512 ? T / 16 : 512) : 1;   // ~16 rows per thread: 2048 workgroups at T = 8192
    if (shard)
      hipLaunchKernelGGL((prep_colmax_kernel<DT, false>), dim3((d / 4 + 255) / 256, ychunks), dim3(256), 0, s, x, b_dec, T, d,
                         (float *)nullptr, colmax);
    else
      hipLaunchKernelGGL((prep_colmax_kernel<DT, true>), dim3((d / 4 + 255) / 256, ychunks), dim3(256), 0, s, x, b_dec, T, d, a32,
                         colmax);
    hipLaunchKernelGGL(pick_outliers_kernel, dim3(1), dim3(1024), 0, s, colmax, d, odims, is_out);
    const unsigned need = skinny ? (PREP_I8 | PREP_FRAG) : PREP_I8;   // operands this call's candidate passes read
    // the candidate passes (gemm_mfma.h, gemm_skinny.h) subtract the shared dither again: both roundings uniform with variance 1/12, for every input
    sd = co.seed != 0ull && getenv("MSAE_NO_SUBTRACT") == nullptr;
    int2 *rowe = reinterpret_cast<int2 *>(ws + pl.off_rowe);
    const unsigned long long *dseed_p = reinterpret_cast<const unsigned long long *>(prepared + offsetof(Prepared, dseed));
    const int *sdtab = reinterpret_cast<const int *>(prepared + pp.off_sdtab);
    if (sd) { zz12 = z * z / 12.f * sd_slack(z, d); zzx = zz12; }
    if (shard) {  // no re-score on this rank: quantise straight from x - b_dec, a32 is never written
      if (sd)
        hipLaunchKernelGGL((quant_x_kernel<DT, true, true>), dim3(pl.Tp), dim3(256), 0, s, x, b_dec, T, d, odims, is_out, xq, xqo, rowc, zz12,
                           tile_major, valid, need, co.seed, dseed_p, rowe, sdtab);
      else
        hipLaunchKernelGGL((quant_x_kernel<DT, true>), dim3(pl.Tp), dim3(256), 0, s, x, b_dec, T, d, odims, is_out, xq, xqo, rowc, zz12, tile_major,
                           valid, need, co.seed, dseed_p, rowe);
    } else if (sd)   // (from a32: reading the caller's 16-bit x + b_dec instead measured +0.016 ms -- the kernel is bound by its instructions, not its bytes)
      hipLaunchKernelGGL((quant_x_kernel<MSAE_F32, false, true>), dim3(pl.Tp), dim3(256), 0, s, (const void *)a32, (const float *)nullptr,
                         T, d, odims, is_out, xq, xqo, rowc, zz12, tile_major, valid, need, co.seed, dseed_p, rowe, sdtab);
    else
      hipLaunchKernelGGL((quant_x_kernel<MSAE_F32, false>), dim3(pl.Tp), dim3(256), 0, s, (const void *)a32, (const float *)nullptr,
                         T, d, odims, is_out, xq, xqo, rowc, zz12, tile_major, valid, need, co.seed, dseed_p, rowe);
    skip_sample = MAIN_SKIPS_SAMPLE && w_packed;   // the tile-major main operand holds the non-sample rows only
    cc_perm = reinterpret_cast<f32x4 *>(ws + pl.off_colc_p);
    hipLaunchKernelGGL(gather_wo_kernel, dim3(N / 32), dim3(256), 0, s, wq, N, d, odims,
                       reinterpret_cast<const f32x4 *>(prepared + pp.off_wstat), wqo, wqos, cc_main, cc_samp, cc_perm,
                       skip_sample ? 1 : 0, sd ? reinterpret_cast<const float *>(prepared + pp.off_ds) : (const float *)nullptr,
                       reinterpret_cast<float *>(ws + pl.off_cds), reinterpret_cast<float *>(ws + pl.off_cds_s),
                       reinterpret_cast<float *>(ws + pl.off_cds_p));
    colc = cc_main; colc_s = cc_samp;
    op_main.A = reinterpret_cast<const unsigned char *>(xq); op_main.ldA = d;
    op_main.B = skinny ? prepared + pp.off_wqf : tile_major ? prepared + pp.off_wqp : reinterpret_cast<const unsigned char *>(wq);
    op_main.ldB = d;
    op_main.nk = tile_major == 2 ? d / 64 : d / 128;
    op_main.packed = skinny ? 3 : tile_major;
    op_main.Ao = reinterpret_cast<const unsigned char *>(xqo);
    op_main.Bo = reinterpret_cast<const unsigned char *>(wqo);
    op_main.n_out = odims + MAX_OUT;
    op_samp = op_main;
    op_samp.B = skinny ? prepared + pp.off_wqsf : tile_major ? prepared + pp.off_wqsp : reinterpret_cast<const unsigned char *>(wqs);
    op_samp.Bo = reinterpret_cast<const unsigned char *>(wqos);
    if constexpr (msae_tuning::ABL_NOLEAD) { op_main.Ao = nullptr; op_main.Bo = nullptr; }   // (tuning builds; results invalid)
  } else if (pl.f8) {
    // e4m3 operands: x scaled per token, W per feature (prepared), both tile-major like the int8 operands; no outlier tile (the
    // format's own dynamic range takes the massive-activation dims), the main pass over ALL features like the bf16 pass
    signed char *x8 = reinterpret_cast<signed char *>(xb);
    unsigned *colmax = reinterpret_cast<unsigned *>(ws + pl.off_colmax);
    int *odims = reinterpret_cast<int *>(ws + pl.off_odims);
    unsigned char *is_out = ws + pl.off_isout;
    f32x4 *cc_main = reinterpret_cast<f32x4 *>(ws + pl.off_colc), *cc_samp = reinterpret_cast<f32x4 *>(ws + pl.off_colc_s);
    const int ychunks = T >= 32 ? (T / 16 < 512 ? T / 16 : 512) : 1;
    hipLaunchKernelGGL((prep_colmax_kernel<DT, true>), dim3((d / 4 + 255) / 256, ychunks), dim3(256), 0, s, x, b_dec, T, d, a32,
                       colmax);
    hipLaunchKernelGGL(pick_outliers_kernel, dim3(1), dim3(1024), 0, s, colmax, d, odims, is_out);
    hipLaunchKernelGGL(quant_x_fp8_kernel, dim3(pl.Tp), dim3(256), 0, s, (const float *)a32, T, d, (const unsigned char *)is_out, x8,
                       rowc, z * z, valid);
    hipLaunchKernelGGL(gather_wo_fp8_kernel, dim3(N / 32), dim3(256), 0, s, W_enc, N, d, (const int *)odims,
                       reinterpret_cast<const f32x4 *>(prepared + pp.off_colbf), cc_main, cc_samp);
    colc = cc_main; colc_s = cc_samp;
    op_main.A = reinterpret_cast<const unsigned char *>(x8); op_main.ldA = d;
    op_main.B = prepared + pp.off_wq; op_main.ldB = d;
    op_main.nk = d / 128;
    op_main.packed = 1;
    op_samp = op_main;
    op_samp.B = prepared + pp.off_wqsp;
  } else {
    hipLaunchKernelGGL(row_p4_kernel, dim3(T), dim3(256), 0, s, a32, T, d, rowc, z * z, valid);
    colc = reinterpret_cast<const f32x4 *>(prepared + pp.off_colbf);
    colc_s = reinterpret_cast<const f32x4 *>(prepared + pp.off_colbf_s);
    op_main.A = reinterpret_cast<const unsigned char *>(xb); op_main.ldA = (size_t)d * 2;
    op_main.B = reinterpret_cast<const unsigned char *>(wb); op_main.ldB = (size_t)d * 2;
    op_main.nk = d / 64;
    op_samp = op_main;
    op_samp.B = reinterpret_cast<const unsigned char *>(wsamp);
  }

  float *refs = reinterpret_cast<float *>(ws + pl.off_refs);
  hipLaunchKernelGGL(band_refs_kernel, dim3(1), dim3(1024), 0, s, colc_s, pl.S, refs);
  prof_mark(co.prof, 1, s);
  {  // sample pass -> dense [T][S]
    GemmEpilogue ep{};
    ep.bias = b_enc; ep.bias_stride = SAMPLE_STRIDE; ep.bias_off = SAMPLE_OFF;
    ep.dense = sample; ep.ld_dense = pl.S;
    ep.rowc = rowc; ep.colc = colc_s; ep.refs = refs; ep.zz12 = zzx;
    if (sd) { ep.row_e = reinterpret_cast<const int2 *>(ws + pl.off_rowe); ep.col_ds = reinterpret_cast<const float *>(ws + pl.off_cds_s); }
    const int grc = skinny == 64    ? gemm_skinny_launch<64, true>(op_samp, T, d, pl.S, ep, s)
                    : skinny == 128 ? gemm_skinny_launch<128, true>(op_samp, T, d, pl.S, ep, s)
                    : skinny == 256 ? gemm_skinny_launch<256, true>(op_samp, T, d, pl.S, ep, s)
                    : pl.i8         ? gemm_launch<GemmI8, true>(op_samp, T, pl.Tp, pl.S, ep, s)
                    : pl.f8         ? gemm_launch<GemmF8, true>(op_samp, T, pl.Tp, pl.S, ep, s)
                                    : gemm_launch<GemmBf16, true>(op_samp, T, pl.Tp, pl.S, ep, s);
    if (grc) return grc;
  }
  prof_mark(co.prof, 2, s);
  int rc = 0;
  // the sample features' own candidates, when the main pass leaves them out: from the threshold select itself (it holds
  // the row in registers), or by sample_push_kernel for the shapes / calls it does not cover (hook edits: features to skip)
  const int skip_a = set_feature >= 0 ? set_feature : -1, skip_b = zero_feature >= 0 ? zero_feature : -1;
  KthPush push{};
  bool pushed = false;
  if (skip_sample && skip_a < 0 && skip_b < 0) {
    push.cnt = pcnt; push.cand = pcand; push.cap = seg_cap; push.stride = SAMPLE_STRIDE; push.off = SAMPLE_OFF;
    push.cnt_stride = pl.segs; push.row_stride = pl.cap;
    pushed = true;
  }
  if (!msae_kth_value_launch(sample, T, pl.S, pl.S, pl.r, tauv, pl.r, pl.r - 1, s, push)) {
    rc = msae_topk_launch(sample, T, pl.S, pl.r, pl.S, nullptr, tauv, taui, s);  // generic shapes
    if (rc) return rc;
    pushed = false;
  }
  if (skip_sample && !pushed)
    hipLaunchKernelGGL(sample_push_kernel, dim3(T), dim3(256), 0, s, sample, pl.S, tauv, pl.r, pl.r - 1, skip_a, skip_b, pcnt,
                       pcand, seg_cap, pl.segs, pl.cap);
  prof_mark(co.prof, 3, s);
  const int N_main = skip_sample ? N - pl.S : N;
  {  // full pass with the threshold epilogue
    GemmEpilogue ep{};
    ep.bias = b_enc; ep.bias_stride = 1; ep.bias_off = 0;
    if (skip_sample) { ep.skip_stride = SAMPLE_STRIDE; ep.skip_off = SAMPLE_OFF; }
    ep.tau_vals = tauv; ep.tau_ld = pl.r; ep.tau_col = pl.r - 1;
    ep.cnt = pcnt; ep.cand = pcand; ep.cap = pl.cap; ep.segs = pl.segs;
    ep.skip_a = set_feature >= 0 ? set_feature : -1;
    ep.skip_b = zero_feature >= 0 ? zero_feature : -1;
    ep.rowc = rowc; ep.colc = skip_sample ? cc_perm : colc; ep.refs = refs; ep.zz12 = zzx;
    if (sd) {
      ep.row_e = reinterpret_cast<const int2 *>(ws + pl.off_rowe);
      ep.col_ds = reinterpret_cast<const float *>(ws + (skip_sample ? pl.off_cds_p : pl.off_cds));
    }
    if constexpr (msae_tuning::GEMM_TIMELINE != 0) {
      if (!g_timeline) (void)hipMalloc(&g_timeline, 64 * 8 * 8);
      (void)hipMemsetAsync(g_timeline, 0, 64 * 8 * 8, s);
      ep.timeline = g_timeline;
    }
    const int grc = skinny == 64    ? gemm_skinny_launch<64, false>(op_main, T, d, N_main, ep, s)
                    : skinny == 128 ? gemm_skinny_launch<128, false>(op_main, T, d, N_main, ep, s)
                    : skinny == 256 ? gemm_skinny_launch<256, false>(op_main, T, d, N_main, ep, s)
                    : pl.i8         ? gemm_launch<GemmI8, false>(op_main, T, pl.Tp, N_main, ep, s)
                    : pl.f8         ? gemm_launch<GemmF8, false>(op_main, T, pl.Tp, N, ep, s)
                                    : gemm_launch<GemmBf16, false>(op_main, T, pl.Tp, N, ep, s);
    if (grc) return grc;
  }
  if (pl.segs > 1)
    hipLaunchKernelGGL(compact_candidates_kernel, dim3(T), dim3(64), 0, s, pcnt, pcand, pl.segs, pl.cap, cnt, cand);
  prof_mark(co.prof, 4, s);
  if (shard) {   // feature-sharded group: this shard's best candidates travel, the owner of the token re-scores
    PackArgs pa{};
    pa.cnt = cnt; pa.cand = cand; pa.cap = pl.cap;
    pa.tau_vals = tauv; pa.tau_ld = pl.r; pa.tau_col = pl.r - 1;
    pa.rowc = rowc; pa.colc = colc; pa.zz12 = zzx; pa.i8 = (pl.i8 || pl.f8) ? 1 : 0;
    pa.C = shard->C; pa.row_offset = shard->row_offset; pa.stride = shard_record_bytes(shard->C);
    pa.recs = shard->recs;
    if (pl.cap <= 64 * 32) hipLaunchKernelGGL(pack_candidates_kernel<32>, dim3(T), dim3(64), 0, s, pa);
    else if (pl.cap <= 64 * 64) hipLaunchKernelGGL(pack_candidates_kernel<64>, dim3(T), dim3(64), 0, s, pa);
    else return MSAE_ENOTIMPL;
    prof_mark(co.prof, 5, s);
    prof_mark(co.prof, 6, s);
    prof_step(co.prof);
    return msae_launch_status();
  }
  {
    RescoreArgs ra{};
    ra.a32 = a32; ra.W_enc = W_enc; ra.b_enc = b_enc;
    ra.tau_vals = tauv; ra.tau_ld = pl.r; ra.tau_col = pl.r - 1;
    ra.cnt = cnt; ra.cand = cand; ra.cap = pl.cap;
    ra.T = T; ra.d = d; ra.N = N; ra.k = k; ra.r_max = pl.r_max;
    ra.rowc = rowc; ra.colc = colc; ra.zz12 = zzx; ra.z2 = z * z; ra.i8 = (pl.i8 || pl.f8) ? 1 : 0;   // (fp8: the three-term band as well)
    // (subtractive dither: the band's sigma is the residuals' actual one again -- 6 sigma, as for round to nearest)
    ra.zc2 = guard_z_check2(pl.i8 && co.seed != 0ull && !sd);
    ra.set_feature = set_feature; ra.set_value = set_value; ra.zero_feature = zero_feature;
    ra.vals = vals; ra.idx = idx.i32; ra.idx64 = idx.i64; ra.status = status; ra.flagged = flagged; ra.n_flagged = n_flagged;
    ra.fb_cap = T;
    ra.rows_out = co.rows_out;
    const int lrc = rescore_stage<DT>(ra, pl, x, b_dec, W_enc, b_enc, a32, cand, ws, T, d, N, k, s);
    if (lrc) return lrc;
  }
  prof_mark(co.prof, 5, s);
  if constexpr (!msae_tuning::ABL_NOFALLBACK) {
    rc = run_exact_fallback<DT>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx,
                                status, ws, pl, co.detail, s);
    if (rc) return rc;
  }
  prof_mark(co.prof, 6, s);
  prof_step(co.prof);
  return msae_launch_status();
}

// msae_options::certified: the same pipeline as run_fast's int8 branch with the certified candidate pass (encode_cert.h) in
// front of the unchanged select + exact re-score -- two planes per operand, no outlier tile, static column constants, z = 1 (the
// band IS the bound) and a model check at exactly the band (a violation can only mean operands that do not belong to the weights).
template <int DT>
int run_cert(const void *x, const float *W_enc, const float *b_enc, const float *b_dec, const unsigned char *cprep, int T, int d,
             int N, int k, int set_feature, float set_value, int zero_feature, float *vals, IdxOut idx, int32_t *status,
             unsigned char *ws, const FusedPlan &pl, const CallOpts &co, hipStream_t s) {
  const CertPrepared cp = make_cert_prepared(N, d);
  float *a32 = reinterpret_cast<float *>(ws + pl.off_a32);
  float *sample = reinterpret_cast<float *>(ws + pl.off_sample);
  float *tauv = reinterpret_cast<float *>(ws + pl.off_tauv);
  int32_t *taui = reinterpret_cast<int32_t *>(ws + pl.off_taui);
  int *cnt = reinterpret_cast<int *>(ws + pl.off_cnt);
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + pl.off_cand);
  int *flagged = reinterpret_cast<int *>(ws + pl.off_flag);
  int *n_flagged = flagged + T;
  signed char *xp = reinterpret_cast<signed char *>(ws + pl.off_xq);
  f32x4 *rowc = reinterpret_cast<f32x4 *>(ws + pl.off_rowc);
  float *refs = reinterpret_cast<float *>(ws + pl.off_refs);
  const float *b_up = reinterpret_cast<const float *>(cprep + cp.off_bup);
  const f32x4 *colc = reinterpret_cast<const f32x4 *>(cprep + cp.off_colc);
  const f32x4 *colc_p = reinterpret_cast<const f32x4 *>(cprep + cp.off_colc_p);
  const f32x4 *colc_s = reinterpret_cast<const f32x4 *>(cprep + cp.off_colc_s);
  (void)b_enc;
  prof_mark(co.prof, 0, s);
  int *pcnt = pl.segs > 1 ? reinterpret_cast<int *>(ws + pl.off_segcnt) : cnt;
  unsigned long long *pcand = pl.segs > 1 ? reinterpret_cast<unsigned long long *>(ws + pl.off_segcand) : cand;
  const int seg_cap = pl.cap / pl.segs;
  const size_t n_cnt = pl.segs > 1 ? (pl.off_segcnt - pl.off_cnt) / 4 + (size_t)T * pl.segs : (size_t)T;
  hipLaunchKernelGGL(zero3_i32_kernel, dim3(64), dim3(256), 0, s, cnt, n_cnt, flagged, (size_t)T + 64 + pl.fb_chunks,
                     (int *)nullptr, (size_t)0);
  hipLaunchKernelGGL(prep_x_kernel<DT>, dim3(2048), dim3(256), 0, s, x, b_dec, T, T, d, (unsigned short *)nullptr, a32);
  hipLaunchKernelGGL(cert_quant_x_kernel, dim3(pl.Tp), dim3(256), 0, s, (const float *)a32, T, d, xp, rowc,
                     reinterpret_cast<const unsigned *>(cprep), N);
  hipLaunchKernelGGL(band_refs_kernel, dim3(1), dim3(1024), 0, s, colc_s, pl.S, refs);
  GemmOperands op_main{}, op_samp{};
  op_main.A = reinterpret_cast<const unsigned char *>(xp); op_main.ldA = d; op_main.ldB = d;
  op_main.B = cprep + cp.off_w;
  op_main.cert = d / 128; op_main.nk = 3 * op_main.cert; op_main.packed = 1;
  op_samp = op_main;
  op_samp.B = cprep + cp.off_ws;
  prof_mark(co.prof, 1, s);
  {
    GemmEpilogue ep{};
    ep.bias = b_up; ep.bias_stride = SAMPLE_STRIDE; ep.bias_off = SAMPLE_OFF;
    ep.dense = sample; ep.ld_dense = pl.S;
    ep.rowc = rowc; ep.colc = colc_s; ep.refs = refs; ep.zz12 = CERT_ZZX;
    const int grc = gemm_launch<GemmI8Cert, true>(op_samp, T, pl.Tp, pl.S, ep, s);
    if (grc) return grc;
  }
  prof_mark(co.prof, 2, s);
  int rc = 0;
  const int skip_a = set_feature >= 0 ? set_feature : -1, skip_b = zero_feature >= 0 ? zero_feature : -1;
  KthPush push{};
  bool pushed = false;
  if (skip_a < 0 && skip_b < 0) {
    push.cnt = pcnt; push.cand = pcand; push.cap = seg_cap; push.stride = SAMPLE_STRIDE; push.off = SAMPLE_OFF;
    push.cnt_stride = pl.segs; push.row_stride = pl.cap;
    pushed = true;
  }
  if (!msae_kth_value_launch(sample, T, pl.S, pl.S, pl.r, tauv, pl.r, pl.r - 1, s, push)) {
    rc = msae_topk_launch(sample, T, pl.S, pl.r, pl.S, nullptr, tauv, taui, s);
    if (rc) return rc;
    pushed = false;
  }
  if (!pushed)
    hipLaunchKernelGGL(sample_push_kernel, dim3(T), dim3(256), 0, s, sample, pl.S, tauv, pl.r, pl.r - 1, skip_a, skip_b, pcnt,
                       pcand, seg_cap, pl.segs, pl.cap);
  prof_mark(co.prof, 3, s);
  {
    GemmEpilogue ep{};
    ep.bias = b_up; ep.bias_stride = 1; ep.bias_off = 0;
    ep.skip_stride = SAMPLE_STRIDE; ep.skip_off = SAMPLE_OFF;
    ep.tau_vals = tauv; ep.tau_ld = pl.r; ep.tau_col = pl.r - 1;
    ep.cnt = pcnt; ep.cand = pcand; ep.cap = pl.cap; ep.segs = pl.segs;
    ep.skip_a = skip_a; ep.skip_b = skip_b;
    ep.rowc = rowc; ep.colc = colc_p; ep.refs = refs; ep.zz12 = CERT_ZZX;
    const int grc = gemm_launch<GemmI8Cert, false>(op_main, T, pl.Tp, N - pl.S, ep, s);
    if (grc) return grc;
  }
  if (pl.segs > 1)
    hipLaunchKernelGGL(compact_candidates_kernel, dim3(T), dim3(64), 0, s, pcnt, pcand, pl.segs, pl.cap, cnt, cand);
  prof_mark(co.prof, 4, s);
  {
    RescoreArgs ra{};
    ra.a32 = a32; ra.W_enc = W_enc; ra.b_enc = b_enc;
    ra.tau_vals = tauv; ra.tau_ld = pl.r; ra.tau_col = pl.r - 1;
    ra.cnt = cnt; ra.cand = cand; ra.cap = pl.cap;
    ra.T = T; ra.d = d; ra.N = N; ra.k = k; ra.r_max = pl.r_max;
    ra.rowc = rowc; ra.colc = colc; ra.zz12 = CERT_ZZX; ra.z2 = 1.f; ra.i8 = 1;
    ra.zc2 = 1.f;                       // |p - c| <= band, always: the check can only trip on foreign operands
    ra.set_feature = set_feature; ra.set_value = set_value; ra.zero_feature = zero_feature;
    ra.vals = vals; ra.idx = idx.i32; ra.idx64 = idx.i64; ra.status = status; ra.flagged = flagged; ra.n_flagged = n_flagged;
    ra.fb_cap = T;
    ra.rows_out = co.rows_out;
    const int lrc = rescore_stage<DT>(ra, pl, x, b_dec, W_enc, b_enc, a32, cand, ws, T, d, N, k, s);
    if (lrc) return lrc;
  }
  prof_mark(co.prof, 5, s);
  rc = run_exact_fallback<DT>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, ws,
                              pl, co.detail, s);
  if (rc) return rc;
  prof_mark(co.prof, 6, s);
  prof_step(co.prof);
  return msae_launch_status();
}

}  // namespace

#ifdef MSAE_GEMM_TIMELINE   // entry points of instrumented builds only (tools/gemm_timeline.py, tools/rescore_timeline.py): not in include/msae.h
extern "C" int msae_debug_timeline(unsigned long long *host_out) {
  if (!g_timeline) return MSAE_EINVAL;
  return (int)hipMemcpy(host_out, g_timeline, 64 * 8 * 8, hipMemcpyDeviceToHost);
}
#endif

#ifdef MSAE_RESCORE_TL
extern "C" int msae_debug_rescore_timeline(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rs_tl), 64 * 16 * 8);
}
#endif

extern "C" void msae_options_init(msae_options *opts) {
  if (!opts) return;
  opts->size = (uint32_t)sizeof(msae_options);
  opts->coarse_mode = MSAE_COARSE_DEFAULT;
  opts->guard_z = 0.f;
  opts->status_detail = 0;
  opts->profile = nullptr;
  opts->exact = 0;
  opts->dither = MSAE_DITHER_DEFAULT;
  opts->rows_rescored = nullptr;
  opts->dither_seed = 0;
  opts->certified = 0;
  opts->reserved2 = 0;
  opts->certified_operands = nullptr;
}

extern "C" int msae_profile_create(int max_steps, void **handle) {
  if (max_steps <= 0 || max_steps > 4096 || !handle) return MSAE_EINVAL;
  ProfState *pf = new (std::nothrow) ProfState();       // (no exception may cross the C ABI)
  if (!pf) return (int)hipErrorOutOfMemory;
  pf->ev = new (std::nothrow) hipEvent_t[(size_t)max_steps * PROF_MARKS];
  if (!pf->ev) { delete pf; return (int)hipErrorOutOfMemory; }
  for (int i = 0; i < max_steps * PROF_MARKS; ++i) {
    const hipError_t e = hipEventCreate(&pf->ev[i]);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(pf->ev[j]);
      delete[] pf->ev;
      delete pf;
      return (int)e;
    }
  }
  pf->max_steps = max_steps;
  *handle = pf;
  return 0;
}

extern "C" int msae_profile_read(void *handle, float *stage_ms, int *n_steps) {
  ProfState *pf = static_cast<ProfState *>(handle);
  if (!pf || pf->magic != 0x50524F46u || !stage_ms) return MSAE_EINVAL;
  const int n = pf->step;
  if (n_steps) *n_steps = n;
  for (int st = 0; st < n; ++st) {
    MSAE_HIP_TRY(hipEventSynchronize(pf->ev[st * PROF_MARKS + PROF_MARKS - 1]));
    for (int i = 0; i + 1 < PROF_MARKS; ++i)
      MSAE_HIP_TRY(hipEventElapsedTime(&stage_ms[st * (PROF_MARKS - 1) + i], pf->ev[st * PROF_MARKS + i],
                                       pf->ev[st * PROF_MARKS + i + 1]));
  }
  pf->step = 0;
  return 0;
}

extern "C" int msae_profile_destroy(void *handle) {
  ProfState *pf = static_cast<ProfState *>(handle);
  if (!pf || pf->magic != 0x50524F46u) return MSAE_EINVAL;
  for (int i = 0; i < pf->max_steps * PROF_MARKS; ++i) (void)hipEventDestroy(pf->ev[i]);
  delete[] pf->ev;
  pf->magic = 0;
  delete pf;
  return 0;
}

extern "C" size_t msae_encoder_prepared_bytes(int N, int d) {
  if (N <= 0 || d <= 0) return 0;
  return make_prepared(N, d).bytes;
}

namespace {
// modes: bit 0 = bf16 operands, bit 1 = int8 operands, bit 2 = without the fragment-major copies (the weight-stream kernels of
// batches of <= 128 tokens read them; the caller refreshes for a large batch)
int prepare_impl(const float *W_enc, int N, int d, void *prepared, int modes, unsigned long long seed, hipStream_t s) {
  if (N <= 0 || d <= 0 || !prepared) return MSAE_EINVAL;
  if (!msae_aligned(prepared, 256)) return MSAE_EALIGN;
  Prepared p = make_prepared(N, d);
  // what this call rebuilds is valid, everything else is stale from now on (the weights have changed)
  p.valid = prep_valid_bits(modes, N, d);
  p.dseed = ((modes & 2) && !(modes & 8) && i8_shape_ok(N, d)) ? seed : 0ull;   // the int8 operands' shared dither (encode_defs.h)
  MSAE_HIP_TRY(hipMemcpyAsync(prepared, &p, sizeof(p), hipMemcpyHostToDevice, s));
  if (p.S) {
    if (!msae_aligned(W_enc, 16)) return MSAE_EALIGN;
    unsigned char *base = static_cast<unsigned char *>(prepared);
    if (modes & 1)
      hipLaunchKernelGGL(prepare_weights_kernel, dim3(4096), dim3(256), 0, s, W_enc, N, d,
                         reinterpret_cast<unsigned short *>(base + p.off_wb),
                         reinterpret_cast<unsigned short *>(base + p.off_ws));
    const bool i8 = i8_shape_ok(N, d);
    RowQuantOut ro = row_quant_out(base, p, modes, i8);
    ro.seed = seed;
    if (p.dseed != 0ull)
      hipLaunchKernelGGL(sd_table_kernel, dim3(1), dim3(1024), 0, s, seed, d, reinterpret_cast<int *>(base + p.off_sdtab));
    if ((modes & 2) && i8 && !(modes & 8))   // row statistics (both passes' error bands) + int8 operands
      hipLaunchKernelGGL(row_stats_quant_kernel<true>, dim3(N), dim3(256), 0, s, W_enc, N, d, ro);
    else
      hipLaunchKernelGGL(row_stats_quant_kernel<false>, dim3(N), dim3(256), 0, s, W_enc, N, d, ro);
    if ((modes & 8) && i8)                    // fp8 operands where the int8 ones would be (PREP_F8)
      hipLaunchKernelGGL(quant_w_fp8_kernel, dim3(N), dim3(256), 0, s, W_enc, N, d,
                         reinterpret_cast<const f32x4 *>(base + p.off_colbf), reinterpret_cast<signed char *>(base + p.off_wq),
                         reinterpret_cast<signed char *>(base + p.off_wqsp));
  }
  return msae_launch_status();
}
}  // namespace

extern "C" int msae_encoder_prepare_opts(const float *W_enc, int N, int d, void *prepared, const msae_options *opts,
                                         void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  // (fp8: its operands take the int8 operands' place -- bf16 + fp8; every other mode: bf16 + int8, either pass can run)
  return prepare_impl(W_enc, N, d, prepared, co.mode == 2 ? (1 | 8) : 3, co.seed, (hipStream_t)stream);
}
extern "C" int msae_encoder_prepare(const float *W_enc, int N, int d, void *prepared, void *stream) {
  return msae_encoder_prepare_opts(W_enc, N, d, prepared, nullptr, stream);
}

// After a weight update (training): rebuild only the operands the coarse mode in force reads.
extern "C" int msae_encoder_refresh(const float *W_enc, int N, int d, void *prepared, const msae_options *opts,
                                    void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  const bool i8 = co.mode == 1 && i8_shape_ok(N, d);
  if (co.mode == 2) return prepare_impl(W_enc, N, d, prepared, 1 | 8, co.seed, (hipStream_t)stream);
  return prepare_impl(W_enc, N, d, prepared, i8 ? 2 : 1, co.seed, (hipStream_t)stream);
}

// ... for an encode of T_next tokens that follows: a batch of more than 256 tokens does not read the fragment-major copies (0.5 GB
// of scattered 16-byte stores per refresh at C2).  The buffer must be refreshed again before an encode of fewer tokens.
extern "C" int msae_encoder_refresh_for(const float *W_enc, int N, int d, void *prepared, int T_next, const msae_options *opts,
                                        void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co) || T_next <= 0) return MSAE_EINVAL;
  const bool i8 = co.mode == 1 && i8_shape_ok(N, d);
  if (co.mode == 2) return prepare_impl(W_enc, N, d, prepared, 1 | 8, co.seed, (hipStream_t)stream);
  return prepare_impl(W_enc, N, d, prepared, (i8 ? 2 : 1) | (T_next > 256 ? 4 : 0), co.seed, (hipStream_t)stream);
}

extern "C" size_t msae_encode_topk_ws_bytes(int T, int d, int N, int k, const msae_options *opts) {
  if (T <= 0 || d <= 0 || N <= 0 || k <= 0) return 0;
  CallOpts co;
  if (!resolve_opts(opts, co)) return 0;
  const FusedPlan pl = make_plan(T, d, N, k, co.mode, 0, co.cert != 0);
  return pl.bytes;
}

extern "C" size_t msae_encoder_certified_bytes(int N, int d) {
  if (N <= 0 || d <= 0 || !cert_shape_ok(N, d)) return 0;
  return make_cert_prepared(N, d).bytes;
}

extern "C" int msae_encoder_prepare_certified(const float *W_enc, const float *b_enc, int N, int d, void *operands, void *stream) {
  if (N <= 0 || d <= 0 || !operands || !W_enc) return MSAE_EINVAL;
  if (!cert_shape_ok(N, d)) return MSAE_ENOTIMPL;
  if (!msae_aligned(operands, 256) || !msae_aligned(W_enc, 16)) return MSAE_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const CertPrepared cp = make_cert_prepared(N, d);
  MSAE_HIP_TRY(hipMemcpyAsync(operands, &cp, sizeof(cp), hipMemcpyHostToDevice, s));
  unsigned char *base = static_cast<unsigned char *>(operands);
  hipLaunchKernelGGL(cert_prepare_rows_kernel, dim3(N), dim3(256), 0, s, W_enc, b_enc, N, d,
                     reinterpret_cast<float *>(base + cp.off_bup), reinterpret_cast<f32x4 *>(base + cp.off_colc),
                     reinterpret_cast<f32x4 *>(base + cp.off_colc_p), reinterpret_cast<f32x4 *>(base + cp.off_colc_s),
                     reinterpret_cast<signed char *>(base + cp.off_w), reinterpret_cast<signed char *>(base + cp.off_ws));
  return msae_launch_status();
}

static int encode_topk_impl(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                            const float *b_dec, const void *prepared, int T, int d, int N, int k,
                            int set_feature, float set_value, int zero_feature, float *vals,
                            IdxOut idx, int32_t *status, void *ws, size_t ws_bytes, const msae_options *opts,
                            void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  if (T < 0 || d <= 0 || N <= 0 || k <= 0 || k > N || k > 4096) return MSAE_EINVAL;
  if (x_dtype != MSAE_F32 && x_dtype != MSAE_BF16 && x_dtype != MSAE_F16) return MSAE_EINVAL;
  if (set_feature >= N || zero_feature >= N) return MSAE_EINVAL;
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  FusedPlan pl = make_plan(T, d, N, k, co.mode, 0, co.cert != 0);
  if (co.cert) {
    if (pl.fast && !co.cert_ops) return MSAE_EINVAL;   // msae_options::certified needs msae_encoder_prepare_certified()'s buffer
  } else if (!prepared && pl.fast) return MSAE_EINVAL;  // the fast path needs msae_encoder_prepare()
  if (ws_bytes < pl.bytes || !ws) return MSAE_EWS;
  if (!msae_aligned(ws, 256)) return MSAE_EALIGN;
  unsigned char *wsb = static_cast<unsigned char *>(ws);
  if (!pl.fast) {
    float *dense = reinterpret_cast<float *>(wsb + pl.off_dense);
    int rc = msae_pre_acts_launch(x, x_dtype, W_enc, b_enc, b_dec, nullptr, nullptr, T, d, N, 1,
                                  dense, N, s);
    if (rc) return rc;
    if (set_feature >= 0 || zero_feature >= 0)
      hipLaunchKernelGGL(edit_dense_kernel, dim3((T + 255) / 256), dim3(256), 0, s, dense, N, T,
                         (const int *)nullptr, set_feature, set_value, zero_feature);
    TopkExtra ex;
    ex.idx64 = idx.i64;
    rc = msae_topk_launch(dense, T, N, k, N, nullptr, vals, idx.i32, s, ex);
    if (rc) return rc;
    if (status) hipLaunchKernelGGL(zero_i32_kernel, dim3(64), dim3(256), 0, s, status, (size_t)T);
    return msae_launch_status();
  }
  const Prepared pp = make_prepared(N, d);  // layout is a pure function of (N, d)
  const unsigned char *pb = static_cast<const unsigned char *>(prepared);
  if (!msae_aligned(x, x_dtype == MSAE_F32 ? 16 : 8) || !msae_aligned(W_enc, 16) ||
      (b_dec && !msae_aligned(b_dec, 16)))
    return MSAE_EALIGN;
  if (co.rows_out && (co.exact || (pl.small && !co.cert)))   // paths without the large-batch re-score kernel report 0 rows
    hipLaunchKernelGGL(zero_i32_kernel, dim3(8), dim3(256), 0, s, co.rows_out, (size_t)T);
  if (co.exact) {   // msae_options::exact: every token through the in-call exact path (bounded scratch, status 1)
    int *flagged = reinterpret_cast<int *>(wsb + pl.off_flag);
    const int n_list = T + 64 + pl.fb_chunks;
    hipLaunchKernelGGL(iota_list_kernel, dim3((n_list + 255) / 256), dim3(256), 0, s, flagged, T, n_list);
    int rc;
    switch (x_dtype) {
      case MSAE_F32: rc = run_exact_fallback<MSAE_F32>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, 0, s); break;
      case MSAE_BF16: rc = run_exact_fallback<MSAE_BF16>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, 0, s); break;
      default: rc = run_exact_fallback<MSAE_F16>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, 0, s); break;
    }
    return rc ? rc : msae_launch_status();
  }
  if (co.cert) {
    const unsigned char *cb = static_cast<const unsigned char *>(co.cert_ops);
    if (!msae_aligned(cb, 256)) return MSAE_EALIGN;
    switch (x_dtype) {
      case MSAE_F32: return run_cert<MSAE_F32>(x, W_enc, b_enc, b_dec, cb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
      case MSAE_BF16: return run_cert<MSAE_BF16>(x, W_enc, b_enc, b_dec, cb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
      default: return run_cert<MSAE_F16>(x, W_enc, b_enc, b_dec, cb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
    }
  }
  if (pl.small) {
    switch (x_dtype) {
      case MSAE_F32: return run_small<MSAE_F32>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
      case MSAE_BF16: return run_small<MSAE_BF16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
      default: return run_small<MSAE_F16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
    }
  }
  switch (x_dtype) {
    case MSAE_F32: return run_fast<MSAE_F32>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
    case MSAE_BF16: return run_fast<MSAE_BF16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
    default: return run_fast<MSAE_F16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, co, s);
  }
}

extern "C" int msae_encode_topk(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                const float *b_dec, const void *prepared, int T, int d, int N, int k,
                                int set_feature, float set_value, int zero_feature, float *vals,
                                int32_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                                const msae_options *opts, void *stream) {
  if (!idx) return MSAE_EINVAL;
  return encode_topk_impl(x, x_dtype, W_enc, b_enc, b_dec, prepared, T, d, N, k, set_feature, set_value,
                          zero_feature, vals, IdxOut{idx, nullptr}, status, ws, ws_bytes, opts, stream);
}

extern "C" int msae_encode_topk_i64(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                    const float *b_dec, const void *prepared, int T, int d, int N, int k,
                                    int set_feature, float set_value, int zero_feature, float *vals,
                                    int64_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                                    const msae_options *opts, void *stream) {
  if (!idx) return MSAE_EINVAL;
  return encode_topk_impl(x, x_dtype, W_enc, b_enc, b_dec, prepared, T, d, N, k, set_feature, set_value,
                          zero_feature, vals, IdxOut{nullptr, idx}, status, ws, ws_bytes, opts, stream);
}

// ---- exact encode of a device-side token list (second round of the feature-sharded engine's per-shard top-k scheme) ----
namespace {
struct RowsPlan { size_t off_counts, off_dense, bytes; int fb_cap, fb_chunks; };
inline RowsPlan make_plan_rows(int max_rows, int N) {
  RowsPlan p{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += msae_align_up(b, 256); return at; };
  p.fb_cap = fallback_capacity(max_rows, N);
  p.fb_chunks = (max_rows + p.fb_cap - 1) / p.fb_cap;
  p.off_counts = take(((size_t)64 + p.fb_chunks) * 4);
  p.off_dense = take((size_t)p.fb_cap * N * 4);
  p.bytes = o;
  return p;
}
}  // namespace

extern "C" size_t msae_encode_topk_rows_ws_bytes(int max_rows, int N) {
  if (max_rows <= 0 || N <= 0) return 0;
  return make_plan_rows(max_rows, N).bytes;
}

extern "C" int msae_encode_topk_rows(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                     const float *b_dec, const int32_t *rows, const int32_t *n_rows, int max_rows,
                                     int d, int N, int k, int set_feature, float set_value, int zero_feature,
                                     float *vals, int64_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                                     void *stream) {
  if (max_rows < 0 || d <= 0 || N <= 0 || k <= 0 || k > N || k > 16384 || !rows || !n_rows || !vals || !idx) return MSAE_EINVAL;
  if (x_dtype != MSAE_F32 && x_dtype != MSAE_BF16 && x_dtype != MSAE_F16) return MSAE_EINVAL;
  if (set_feature >= N || zero_feature >= N) return MSAE_EINVAL;
  if (max_rows == 0) return 0;
  const RowsPlan rp = make_plan_rows(max_rows, N);
  if (ws_bytes < rp.bytes || !ws) return MSAE_EWS;
  if (!msae_aligned(ws, 256) || !msae_aligned(W_enc, 16) || !msae_aligned(x, x_dtype == MSAE_F32 ? 16 : 8) ||
      (b_dec && !msae_aligned(b_dec, 16)))
    return MSAE_EALIGN;
  unsigned char *wsb = static_cast<unsigned char *>(ws);
  int *counts = reinterpret_cast<int *>(wsb + rp.off_counts) + 64;
  float *dense = reinterpret_cast<float *>(wsb + rp.off_dense);
  hipStream_t s = (hipStream_t)stream;
  const IdxOut io{nullptr, idx};
  int rc;
  switch (x_dtype) {
    case MSAE_F32: rc = run_exact_rows<MSAE_F32>(x, W_enc, b_enc, b_dec, rows, n_rows, counts, dense, rp.fb_cap, rp.fb_chunks, d, N, k, set_feature, set_value, zero_feature, vals, io, status, 0, s); break;
    case MSAE_BF16: rc = run_exact_rows<MSAE_BF16>(x, W_enc, b_enc, b_dec, rows, n_rows, counts, dense, rp.fb_cap, rp.fb_chunks, d, N, k, set_feature, set_value, zero_feature, vals, io, status, 0, s); break;
    default: rc = run_exact_rows<MSAE_F16>(x, W_enc, b_enc, b_dec, rows, n_rows, counts, dense, rp.fb_cap, rp.fb_chunks, d, N, k, set_feature, set_value, zero_feature, vals, io, status, 0, s); break;
  }
  return rc ? rc : msae_launch_status();
}

// ---- feature-sharded group: per-shard candidates, owner-side exact re-score ----------------------------
namespace {
struct ExtPlan { size_t off_a32, off_flag, off_fbdense, bytes; int fb_cap, fb_chunks, cap, r_max; };
inline ExtPlan make_plan_ext(int T, int d, int N, int k, int M) {
  ExtPlan p{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += msae_align_up(b, 256); return at; };
  p.cap = next_pow2(M > 2 ? M : 2);
  p.r_max = k <= 64 ? 8 * k : 3 * k;
  if (p.r_max < k + 4) p.r_max = k + 4;
  if (p.r_max > p.cap) p.r_max = p.cap;
  p.off_a32 = take((size_t)T * d * 4);
  p.fb_cap = fallback_capacity(T, N);
  p.fb_chunks = (T + p.fb_cap - 1) / p.fb_cap;
  p.off_flag = take(((size_t)T + 64 + p.fb_chunks) * 4);
  p.off_fbdense = take((size_t)p.fb_cap * N * 4);
  p.bytes = o;
  return p;
}

template <int DT>
int run_rescore_ext(const void *x, const float *W_enc, const float *b_enc, const float *b_dec, int T, int T_valid,
                    int d, int N, int k, int G, int C, const unsigned char *recs, int set_feature, float set_value,
                    int zero_feature, float *vals, int64_t *idx, int32_t *status, unsigned char *ws,
                    const ExtPlan &xp, const CallOpts &co, hipStream_t s) {
  float *a32 = reinterpret_cast<float *>(ws + xp.off_a32);
  int *flagged = reinterpret_cast<int *>(ws + xp.off_flag);
  int *n_flagged = flagged + T;
  const float z = co.z;
  hipLaunchKernelGGL(zero_i32_kernel, dim3(8), dim3(256), 0, s, flagged, (size_t)T + 64 + xp.fb_chunks);
  hipLaunchKernelGGL(prep_x_kernel<DT>, dim3(2048), dim3(256), 0, s, x, b_dec, T_valid, T_valid, d,
                     (unsigned short *)nullptr, a32);
  RescoreArgs ra{};
  ra.a32 = a32; ra.W_enc = W_enc; ra.b_enc = b_enc;
  ra.cap = xp.cap;
  ra.T = T_valid; ra.d = d; ra.N = N; ra.k = k; ra.r_max = xp.r_max;
  ra.zz12 = z * z / 12.f; ra.z2 = z * z; ra.i8 = 0;
  // (the records' z sigma came from shards running with the same options; large batches subtract the dither there -- actual
  // sigma --, small ones carry Hoeffding's proxy: 6 of either is the net under operands edited behind the API)
  ra.zc2 = guard_z_check2(false);
  ra.set_feature = set_feature; ra.set_value = set_value; ra.zero_feature = zero_feature;
  ra.vals = vals; ra.idx = nullptr; ra.idx64 = idx; ra.status = status; ra.flagged = flagged; ra.n_flagged = n_flagged;
  ra.fb_cap = T;
  ra.ext = recs; ra.ext_G = G; ra.ext_C = C; ra.ext_T = T; ra.ext_stride = shard_record_bytes(C); ra.ext_valid = T_valid;
  const int nrp = next_pow2(xp.r_max + 1);
  const size_t smem = ((size_t)xp.cap + nrp) * 8 + (size_t)xp.cap * 8 + 64;
  const int lrc = launch_select_rescore<true>(ra, T_valid, k, smem, (const float *)a32, W_enc, s);
  if (lrc) return lrc;
  FusedPlan pl{};                       // the exact fallback reads only these fields
  pl.off_flag = xp.off_flag; pl.off_fbdense = xp.off_fbdense; pl.fb_cap = xp.fb_cap; pl.fb_chunks = xp.fb_chunks;
  int rc = run_exact_fallback<DT>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals,
                                  IdxOut{nullptr, idx}, status, ws, pl, co.detail, s);
  if (rc) return rc;
  return msae_launch_status();
}
}  // namespace

extern "C" size_t msae_shard_record_bytes(int C) { return C > 0 ? (size_t)shard_record_bytes(C) : 0; }

extern "C" int msae_shard_candidates(const void *x, int x_dtype, const float *b_enc, const float *b_dec,
                                     const void *prepared, int T, int d, int N, int k, int row_offset, int C,
                                     int set_feature, int zero_feature, void *records, void *ws, size_t ws_bytes,
                                     const msae_options *opts, void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  if (T < 0 || d <= 0 || N <= 0 || k <= 0 || C <= 0 || row_offset < 0 || !records) return MSAE_EINVAL;
  if (x_dtype != MSAE_F32 && x_dtype != MSAE_BF16 && x_dtype != MSAE_F16) return MSAE_EINVAL;
  if (T == 0) return 0;
  FusedPlan pl = make_plan(T, d, N, k, co.mode, C);
  if (!pl.fast || !prepared || C > pl.cap || pl.f8) return MSAE_ENOTIMPL;   // shapes / modes without the candidate exchange: msae_encode_topk per shard
  pl.small = false;
  if (ws_bytes < pl.bytes || !ws) return MSAE_EWS;
  if (!msae_aligned(ws, 256) || !msae_aligned(records, 8)) return MSAE_EALIGN;
  if (!msae_aligned(x, x_dtype == MSAE_F32 ? 16 : 8) || (b_dec && !msae_aligned(b_dec, 16))) return MSAE_EALIGN;
  const Prepared pp = make_prepared(N, d);
  const unsigned char *pb = static_cast<const unsigned char *>(prepared);
  unsigned char *wsb = static_cast<unsigned char *>(ws);
  hipStream_t s = (hipStream_t)stream;
  const ShardOut so{static_cast<unsigned char *>(records), C, row_offset};
  // the hooks' features are global ids: only the owning shard leaves them out of its candidates
  const int sf = (set_feature >= row_offset && set_feature < row_offset + N) ? set_feature - row_offset : -1;
  const int zf = (zero_feature >= row_offset && zero_feature < row_offset + N) ? zero_feature - row_offset : -1;
  switch (x_dtype) {
    case MSAE_F32: return run_fast<MSAE_F32>(x, nullptr, b_enc, b_dec, pp, pb, T, d, N, k, sf, 0.f, zf, nullptr, IdxOut{nullptr, nullptr}, nullptr, wsb, pl, co, s, &so);
    case MSAE_BF16: return run_fast<MSAE_BF16>(x, nullptr, b_enc, b_dec, pp, pb, T, d, N, k, sf, 0.f, zf, nullptr, IdxOut{nullptr, nullptr}, nullptr, wsb, pl, co, s, &so);
    default: return run_fast<MSAE_F16>(x, nullptr, b_enc, b_dec, pp, pb, T, d, N, k, sf, 0.f, zf, nullptr, IdxOut{nullptr, nullptr}, nullptr, wsb, pl, co, s, &so);
  }
}

extern "C" size_t msae_rescore_candidates_ws_bytes(int T, int d, int N, int k, int G, int C) {
  if (T <= 0 || d <= 0 || N <= 0 || k <= 0 || G <= 0 || C <= 0) return 0;
  return make_plan_ext(T, d, N, k, G * C).bytes;
}

extern "C" int msae_rescore_candidates(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                       const float *b_dec, int T, int T_valid, int d, int N, int k, int G, int C,
                                       const void *records, int set_feature, float set_value, int zero_feature,
                                       float *vals, int64_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                                       const msae_options *opts, void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  if (T < 0 || T_valid < 0 || T_valid > T || d <= 0 || N <= 0 || k <= 0 || k > N || k > 256 || G <= 0 || C <= 0 ||
      (long)G * C < k || (long)G * C > 8192 || d % 64 != 0)
    return MSAE_EINVAL;
  if (x_dtype != MSAE_F32 && x_dtype != MSAE_BF16 && x_dtype != MSAE_F16) return MSAE_EINVAL;
  if (set_feature >= N || zero_feature >= N || !records || !vals || !idx) return MSAE_EINVAL;
  if (T_valid == 0) return 0;
  const ExtPlan xp = make_plan_ext(T, d, N, k, G * C);
  if (ws_bytes < xp.bytes || !ws) return MSAE_EWS;
  if (!msae_aligned(ws, 256) || !msae_aligned(records, 8) || !msae_aligned(W_enc, 16)) return MSAE_EALIGN;
  if (!msae_aligned(x, x_dtype == MSAE_F32 ? 16 : 8) || (b_dec && !msae_aligned(b_dec, 16))) return MSAE_EALIGN;
  unsigned char *wsb = static_cast<unsigned char *>(ws);
  const unsigned char *rb = static_cast<const unsigned char *>(records);
  hipStream_t s = (hipStream_t)stream;
  switch (x_dtype) {
    case MSAE_F32: return run_rescore_ext<MSAE_F32>(x, W_enc, b_enc, b_dec, T, T_valid, d, N, k, G, C, rb, set_feature, set_value, zero_feature, vals, idx, status, wsb, xp, co, s);
    case MSAE_BF16: return run_rescore_ext<MSAE_BF16>(x, W_enc, b_enc, b_dec, T, T_valid, d, N, k, G, C, rb, set_feature, set_value, zero_feature, vals, idx, status, wsb, xp, co, s);
    default: return run_rescore_ext<MSAE_F16>(x, W_enc, b_enc, b_dec, T, T_valid, d, N, k, G, C, rb, set_feature, set_value, zero_feature, vals, idx, status, wsb, xp, co, s);
  }
}
