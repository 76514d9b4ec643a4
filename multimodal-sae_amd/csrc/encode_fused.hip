// encode_fused.hip -- fused Sae.encode: MFMA candidate pass (int8 or bf16) + exact f32 re-score + TopK.
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
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_mfma.h"
#include "gemm_skinny.h"
#ifdef MSAE_GEMM_RING64   // tuning builds: the 64-byte / 4-slot ring variant of the candidate GEMM (measured slower, kept as a record)
#include "gemm_mfma64.h"
#endif

int msae_pre_acts_launch(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                         const float *b_dec, const int *rows, const int *n_rows, int T, int d, int N,
                         int relu, float *out, int ld_out, hipStream_t s);
int msae_topk_launch(const float *latents, int T, int N, int k, int ld, const int *n_rows,
                     float *vals, int32_t *idx, hipStream_t s, const TopkExtra &ex = TopkExtra());
bool msae_kth_value_launch(const float *rows, int T, int S, int ld, int r, float *out, int out_ld,
                           int out_col, hipStream_t s, const KthPush &push = KthPush());

namespace {

constexpr int SAMPLE_STRIDE = 32, SAMPLE_OFF = 13;
// Tokens one pass of the in-call exact fallback absorbs: its dense scratch rows are budgeted at 1 GiB
// (2048 rows at N = 131072), never fewer than 128 and never more than the call has tokens.  The
// exact kernels take the flagged count from device memory and loop over it; ceil(T / capacity) passes
// are enqueued (the ones without work exit at once), so EVERY flagged token is recomputed inside the
// call whatever their number -- no host round trip, no "unresolved" leftovers.
constexpr size_t FB_BUDGET_BYTES = (size_t)1 << 30;
inline int fallback_capacity(int T, int N) {
  size_t cap = FB_BUDGET_BYTES / ((size_t)N * 4);
  const size_t t128 = ((size_t)T + 127) / 128 * 128;
  if (cap > t128) cap = t128;
  if (cap < 128) cap = 128;
  return (int)(cap / 128 * 128);
}
constexpr int EXACT_T_MAX = 0;      // fused path for every T (T=1: 1 GiB bf16 stream beats the f32 tile 4x)

// ---- prepared encoder ------------------------------------------------------------------------
struct Prepared {
  unsigned magic;
  int N, d, S;
  size_t off_wb, off_ws, off_wstat, off_wstat_s, off_colbf, off_colbf_s, off_wq, off_wqs, off_wqp, off_wqsp, off_wqf, off_wqsf, bytes;
  // Which operand groups hold the CURRENT weights (PREP_* bits).  msae_encoder_refresh[_for] rebuilds only what the following
  // encode reads and clears the bits of everything else; every fused path's prep kernel tests the bits of the operands ITS
  // candidate pass is about to read and, when one is missing, hands all its tokens to the exact path (reason 128) -- stale
  // operands cost time, never a wrong top-k, and nothing about them lives on the host.
  unsigned valid;
};
constexpr unsigned PREP_MAGIC = 0x4D534145u;  // "MSAE"
constexpr unsigned PREP_BF16 = 1u;   // W_bf16 + bf16 sample rows
constexpr unsigned PREP_I8 = 2u;     // Wq row-major, tile-major (+ sample copies)
constexpr unsigned PREP_FRAG = 4u;   // Wq fragment-major (+ sample copy): the weight-stream kernels of <= 128 tokens

__host__ __device__ inline bool fast_shape_ok(int N, int d) {
  return N % (SAMPLE_STRIDE * 256) == 0 && d % 64 == 0;  // sample width N/32 must tile by BN = 256
}
__host__ __device__ inline bool i8_shape_ok(int N, int d) { return fast_shape_ok(N, d) && d % 128 == 0; }

// The tile-major operand of the main candidate pass leaves the sample rows out (the sample pass has scored them: their
// candidates are taken from its output, sample_push_kernel): -1/32 of the pass's matrix work and operand traffic.  Row n of W
// (n not a sample row) is row main_row(n) of that operand; column c of the pass is feature gemm_feature(c).  31/32 N tiles by
// 256 whenever the sample width does (fast_shape_ok).  -DMSAE_FULL_MAIN_PASS (tuning builds) keeps all rows in.
#ifdef MSAE_FULL_MAIN_PASS
constexpr bool MAIN_SKIPS_SAMPLE = false;
#else
constexpr bool MAIN_SKIPS_SAMPLE = true;
#endif
__host__ __device__ inline int main_row(int n) { return n - n / SAMPLE_STRIDE - ((n % SAMPLE_STRIDE) > SAMPLE_OFF ? 1 : 0); }

// 256-B header | W_bf16 [N][d] | sample rows bf16 [S][d] | row statistics (sw, Q_i8, |W_n|^2, Q_bf) f32x4 [N]
// and [S] | bf16-pass column constants (1, Q_bf, 0, 0) f32x4 [N] and [S] | Wq int8 [N][d] | sample int8 [S][d]
// | Wq fragment-major [N/16][d/64][64 lanes][16 B] | sample fragment-major (the weight-stream kernel's operand, gemm_skinny.h)
// | Wq tile-major [N/256][d/128][256][128] | sample tile-major (the candidate GEMM's operands, gemm_mfma.h; the
// row-major copies feed the S = 1 weight streams and the outlier-column gather)
inline Prepared make_prepared(int N, int d) {
  Prepared p{};
  p.magic = PREP_MAGIC;
  p.N = N; p.d = d;
  p.S = fast_shape_ok(N, d) ? N / SAMPLE_STRIDE : 0;
  size_t o = 256;
  auto take = [&](size_t b) { size_t at = o; o += msae_align_up(b, 256); return at; };
  p.off_wb = take(p.S ? (size_t)N * d * 2 : 0);
  p.off_ws = take((size_t)p.S * d * 2);
  const bool q = p.S && i8_shape_ok(N, d);
  p.off_wstat = take(p.S ? (size_t)N * 16 : 0);
  p.off_wstat_s = take((size_t)p.S * 16);
  p.off_colbf = take(p.S ? (size_t)N * 16 : 0);
  p.off_colbf_s = take((size_t)p.S * 16);
  p.off_wq = take(q ? (size_t)N * d : 0);
  p.off_wqs = take(q ? (size_t)p.S * d : 0);
  p.off_wqp = take(q ? (size_t)N * d : 0);
  p.off_wqsp = take(q ? (size_t)p.S * d : 0);
  p.off_wqf = take(q ? (size_t)N * d : 0);
  p.off_wqsf = take(q ? (size_t)p.S * d : 0);
  p.bytes = o;
  return p;
}

// ---- per-call options (msae_options, include/msae.h), resolved once per entry-point call.  The library holds no
// mutable state: the environment only supplies DEFAULTS (read at the call, never cached), everything else travels
// with the call.
struct ProfState;
struct CallOpts {
  int mode;          // coarse-pass operand type: 0 = bf16, 1 = int8
  float z;           // width of the error band: u = coarse + z*sigma
  int detail;        // status = 1 | reason << 8 for tokens recomputed in the call
  ProfState *prof;   // stage timing handle or null
  int exact;         // every token by the exact path (msae_options::exact)
};
inline bool resolve_opts(const msae_options *o, CallOpts &c) {
  c.mode = -1; c.z = 0.f; c.detail = 0; c.prof = nullptr; c.exact = 0;
  if (o) {
    // `size` is the caller's sizeof: a caller compiled against ABI 2's header (no `exact`) is served with exact = 0
    if (o->size < offsetof(msae_options, exact)) return false;
    c.mode = o->coarse_mode; c.z = o->guard_z; c.detail = o->status_detail ? 1 : 0;
    c.prof = static_cast<ProfState *>(o->profile);
    if (o->size >= offsetof(msae_options, exact) + sizeof(int32_t)) c.exact = o->exact ? 1 : 0;
  }
  if (c.mode < 0) {
    const char *e = getenv("MSAE_COARSE");
    c.mode = (e && e[0] == 'b') ? 0 : 1;
  }
  if (c.mode != 0 && c.mode != 1) return false;
  if (c.z == 0.f) {
    const char *e = getenv("MSAE_GUARD_Z");
    const float v = e ? (float)atof(e) : 0.f;
    c.z = (v >= 0.25f && v <= 64.f) ? v : 7.f;
  }
  return c.z >= 0.25f && c.z <= 64.f;
}

#ifdef MSAE_GEMM_TIMELINE
unsigned long long *g_timeline = nullptr;   // tuning builds only
#endif
constexpr float GUARD_Z_CHECK = 6.f;      // a re-scored pair further than this many sigma from its coarse value flags the token
// Deterministic per-token guard of the int8 pass.  The x-side residual is modelled as independent rounding
// noise of variance sx^2 / 12 per dim.  The dims that round to ZERO are the exception: their residual is the
// activation itself, i.e. structured -- a feature whose weights correlate with that part of the token (cosine c) is off by
// up to c sqrt(E0) |W_n|, E0 = their energy, against an x-side band of z sx |W_n| / sqrt(12) = 2.02 sx |W_n| at the default
// z = 7.  For a well-scaled Gaussian token sqrt(E0) = 2.06 sx (one band); a token whose scale is dictated by an isolated
// large dim that the batch-level outlier list did not take rounds most of its dims to zero and sqrt(E0) approaches the
// token's whole norm.  Tokens with sqrt(E0) > GUARD_E0_SX * sx (4 default bands: a feature would need a cosine above
// 0.25 with the zeroed part to leave its band) are not trusted to the statistical model: they are flagged (reason 128)
// and recomputed by the exact path inside the call.  The test does not move with msae_options::guard_z.
constexpr float GUARD_E0_SX = 4.f * 7.f * 0.288675f;   // 4 bands of z = 7: 8.08
#ifndef MSAE_GUARD_ZETA
#define MSAE_GUARD_ZETA 1.f
#endif
constexpr float GUARD_ZETA = MSAE_GUARD_ZETA;   // first round reaches zeta sigma below the k-th coarse value
constexpr float BF16_REL_VAR2 = 5.5e-6f;  // variance of the sum of two relative bf16 roundings (2 x 2^-16/3 x E[1/m^2])

// z^2 sigma^2 of one (token, feature) pair; rc = (sx, m, P, -), cc = (sw, Q, Si, So).  Same expression as the
// GEMM epilogue (gemm_mfma.h).
__device__ __forceinline__ float band_sq(const f32x4 rc, const f32x4 cc, float zz12, bool i8) {
  if (!i8) return rc[2] * cc[1];
  const float rz = rc[0] * rc[0] * zz12;
  return __builtin_fmaf(rc[2], cc[1], __builtin_fmaf(rz * rc[1] * rc[1], cc[3], rz * cc[2]));
}

// W_bf16[n][c] = bf16(W[n][c]); sample row j = row j*SAMPLE_STRIDE + SAMPLE_OFF.  grid-stride over 8-element groups.
__global__ __launch_bounds__(256) void prepare_weights_kernel(const float *__restrict__ W, int N,
                                                              int d, unsigned short *__restrict__ wb,
                                                              unsigned short *__restrict__ ws) {
  const size_t groups = (size_t)N * d / 8;
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
    const size_t e = g * 8;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(W + e);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(W + e + 4);
    u16x8 o;
    o[0] = f32_to_bf16_bits(a[0]); o[1] = f32_to_bf16_bits(a[1]);
    o[2] = f32_to_bf16_bits(a[2]); o[3] = f32_to_bf16_bits(a[3]);
    o[4] = f32_to_bf16_bits(b[0]); o[5] = f32_to_bf16_bits(b[1]);
    o[6] = f32_to_bf16_bits(b[2]); o[7] = f32_to_bf16_bits(b[3]);
    *reinterpret_cast<u16x8 *>(wb + e) = o;
    const size_t n = e / d, c = e % d;
    if (n % SAMPLE_STRIDE == SAMPLE_OFF)
      *reinterpret_cast<u16x8 *>(ws + (n / SAMPLE_STRIDE) * d + c) = o;
  }
}

// a32[t][c] = (float)x[t][c] - b_dec[c] (the exact f32 SAE input, sae.py:174) and
// xb[t][c] = bf16(a32[t][c]) for t < T; xb rows up to Tp are zero.
template <int DT>
__global__ __launch_bounds__(256) void prep_x_kernel(const void *__restrict__ x,
                                                     const float *__restrict__ b_dec, int T, int Tp,
                                                     int d, unsigned short *__restrict__ xb,
                                                     float *__restrict__ a32) {
  const size_t groups = (size_t)Tp * d / 4;
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
    const size_t e = g * 4;
    const size_t t = e / d, c = e % d;
    u16x4 o = {0, 0, 0, 0};
    if ((int)t < T) {
      f32x4 v = load_x4<DT>(x, e);
      if (b_dec) v = v - *reinterpret_cast<const f32x4 *>(b_dec + c);
      *reinterpret_cast<f32x4 *>(a32 + e) = v;
      o[0] = f32_to_bf16_bits(v[0]); o[1] = f32_to_bf16_bits(v[1]);
      o[2] = f32_to_bf16_bits(v[2]); o[3] = f32_to_bf16_bits(v[3]);
    }
    if (xb) *reinterpret_cast<u16x4 *>(xb + e) = o;   // the int8 coarse pass quantises a32 itself
  }
}

// ---- per-row statistics + int8 operands ---------------------------------------------------------------
// Tile-major int8 operand of the candidate GEMM (GemmOperands::packed): byte offset of the 16-B chunk at column c
// (c % 16 == 0) of row r, with the LDS image's chunk permutation applied (gemm_swz).  d % 128 == 0.
// layout 1: 128-byte k-tiles (gemm_mfma.h), layout 2: 64-byte k-tiles (gemm_mfma64.h: packed64_off; tuning builds)
__host__ __device__ __forceinline__ size_t packed_off(size_t r, int c, int d, int layout = 1) {
#ifdef MSAE_GEMM_RING64
  if (layout == 2) return packed64_off(r, c, d);
#endif
  const size_t rt = r >> 8, ri = r & 255;
  const int kt = c >> 7, ch = (c >> 4) & 7;
  return ((rt * (size_t)(d >> 7) + kt) * 256 + ri) * 128 + (size_t)((ch ^ (int)((ri >> 1) & 7)) << 4);
}
// Fragment-major int8 operand of the weight-stream kernel (gemm_skinny.h): the 16 B at column c (c % 16 == 0) of row r sit where
// lane 16 ((c % 64) / 16) + r % 16 of a v_mfma_i32_16x16x64_i8 B fragment reads them -- one k-step of a 16-row block is ONE
// contiguous kilobyte, lane l at byte 16 l.
__host__ __device__ __forceinline__ size_t frag_off(size_t r, int c, int d) {
  return ((((r >> 4) * (size_t)(d >> 6) + (size_t)(c >> 6)) << 6) + (size_t)((((c >> 4) & 3) << 4) + (int)(r & 15))) << 4;
}
// which operand layout the candidate GEMM reads: 1 = tile-major, 128-byte k-tiles in a 2-slot ring (default); 0 =
// row-major (environment MSAE_GEMM_ROWMAJOR=1, for A/B runs); 2 = tile-major 64-byte k-tiles in a 4-slot ring (tuning
// builds with -DMSAE_GEMM_RING64 and MSAE_GEMM_RING64=1 in the environment).  Read at every call: an immutable property
// of the process environment (prepare and encode must agree).
inline int gemm_layout() {
  if (getenv("MSAE_GEMM_ROWMAJOR")) return 0;
#ifdef MSAE_GEMM_RING64
  if (getenv("MSAE_GEMM_RING64")) return 2;
#endif
  return 1;
}

// W side (once per weight load), one 256-thread workgroup per row:
//   sw[n] = max|W[n][:]| / 127,  |W_n|^2,  |W_n|_4^2 = sqrt(sum w^4)   -> wstat[n] = (sw, Q_i8, |W_n|^2, Q_bf)
//   Wq[n][c] = rint(W[n][c] / sw[n])   (QUANT; d % 128 == 0)
// A row whose rms lies below one step (max > 127 rms: its bulk quantises to 0, +-1) would leave a
// STRUCTURED residual (the bulk itself), so such rows are rounded stochastically with a hash dither:
// floor(s + r(n, c)), r uniform in [0, 1) -- unbiased for any activation direction, residual variance
// <= 1/4 step^2 instead of 1/12: Q_i8 = 3 sw^2 for them.
__device__ __forceinline__ float hash01(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(unsigned)(z >> 40) * (1.f / 16777216.f);
}
template <bool QUANT>
__global__ __launch_bounds__(256) void row_stats_quant_kernel(const float *__restrict__ W, int N, int d,
                                                              f32x4 *__restrict__ wstat, f32x4 *__restrict__ wstat_s,
                                                              f32x4 *__restrict__ colbf, f32x4 *__restrict__ colbf_s,
                                                              signed char *__restrict__ wq,
                                                              signed char *__restrict__ wqs,
                                                              signed char *__restrict__ wqp,
                                                              signed char *__restrict__ wqsp,
                                                              signed char *__restrict__ wqf,
                                                              signed char *__restrict__ wqsf, int layout) {
  __shared__ float red[3][4];
  const int n = blockIdx.x;
  const float *row = W + (size_t)n * d;
  float m = 0.f, s2 = 0.f, s4 = 0.f;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float q = v[e] * v[e];
      m = fmaxf(m, fabsf(v[e]));
      s2 += q;
      s4 = __builtin_fmaf(q, q, s4);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m = fmaxf(m, __shfl_xor(m, off, 64));
    s2 += __shfl_xor(s2, off, 64);
    s4 += __shfl_xor(s4, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = s4; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  s4 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  const float scale = m > 0.f ? m / 127.f : 0.f;          // an all-zero row: coarse value = bias exactly, no band
  const bool dither = s2 < scale * scale * (float)d;       // rms below one step
  const bool samp = (n % SAMPLE_STRIDE) == SAMPLE_OFF;
  if (threadIdx.x == 0) {
    const float q_bf = __builtin_sqrtf(s4);
    const f32x4 st = {scale, scale * scale * (dither ? 3.f : 1.f), s2, q_bf};
    const f32x4 cb = {1.f, q_bf, 0.f, 0.f};
    wstat[n] = st;
    colbf[n] = cb;
    if (samp) { wstat_s[n / SAMPLE_STRIDE] = st; colbf_s[n / SAMPLE_STRIDE] = cb; }
  }
  if constexpr (QUANT) {
    const float inv = m > 0.f ? 1.f / scale : 0.f;
    for (int c = threadIdx.x * 16; c < d; c += 4096) {
      i32x4 packed;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c + 4 * q);
        unsigned w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = v[e] * inv;
          int iv = dither ? (int)floorf(sv + hash01((unsigned long long)n * (unsigned)d + (unsigned)(c + 4 * q + e)))
                          : (int)rintf(sv);
          iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
          w |= ((unsigned)iv & 0xFFu) << (8 * e);
        }
        packed[q] = (int)w;
      }
      *reinterpret_cast<i32x4 *>(wq + (size_t)n * d + c) = packed;
      if (layout == 1 && MAIN_SKIPS_SAMPLE) {
        if (!samp) *reinterpret_cast<i32x4 *>(wqp + packed_off((size_t)main_row(n), c, d, 1)) = packed;
      } else {
        *reinterpret_cast<i32x4 *>(wqp + packed_off((size_t)n, c, d, layout)) = packed;
      }
      if (wqf) {                                   // (null: msae_encoder_refresh_for a large batch)
        if (MAIN_SKIPS_SAMPLE) { if (!samp) *reinterpret_cast<i32x4 *>(wqf + frag_off((size_t)main_row(n), c, d)) = packed; }
        else *reinterpret_cast<i32x4 *>(wqf + frag_off((size_t)n, c, d)) = packed;
      }
      if (samp) {
        *reinterpret_cast<i32x4 *>(wqs + (size_t)(n / SAMPLE_STRIDE) * d + c) = packed;
        *reinterpret_cast<i32x4 *>(wqsp + packed_off((size_t)(n / SAMPLE_STRIDE), c, d, layout)) = packed;
        if (wqsf) *reinterpret_cast<i32x4 *>(wqsf + frag_off((size_t)(n / SAMPLE_STRIDE), c, d)) = packed;
      }
    }
  }
}

// x side (every call).  Massive-activation dims would dictate the per-token scale and wipe out
// the resolution of all other dims, so they are split off: colmax -> outlier dim list ->
// per-token quantisation with the outliers in their own 128-wide k-tile at scale m[t]*sx[t].
// int8 pass: prep_x and colmax in one sweep -- a thread owns four columns (b_dec in registers) and walks its
// rows: a32 = x - b_dec is written once and never read back for the maxima.
constexpr int COLMAX_PARTS = 8;   // copies of the column maxima (see prep_colmax_kernel)
template <int DT, bool WRITE_A32>
__global__ __launch_bounds__(256) void prep_colmax_kernel(const void *__restrict__ x, const float *__restrict__ b_dec,
                                                          int T, int d, float *__restrict__ a32,
                                                          unsigned *__restrict__ colmax_bits) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= d) return;
  const int rows_per = (T + gridDim.y - 1) / gridDim.y;
  const int t0 = blockIdx.y * rows_per, t1 = min(T, t0 + rows_per);
  const f32x4 bd = b_dec ? *reinterpret_cast<const f32x4 *>(b_dec + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int t = t0; t < t1; ++t) {
    f32x4 v = load_x4<DT>(x, (size_t)t * d + c);
    if (b_dec) v = v - bd;
    if constexpr (WRITE_A32) *reinterpret_cast<f32x4 *>(a32 + (size_t)t * d + c) = v;
    m[0] = fmaxf(m[0], fabsf(v[0])); m[1] = fmaxf(m[1], fabsf(v[1]));
    m[2] = fmaxf(m[2], fabsf(v[2])); m[3] = fmaxf(m[3], fabsf(v[3]));
  }
  // (blockIdx.y % COLMAX_PARTS: one copy of the maxima for all row chunks means T / 16 atomics on every column's word -- hundreds of
  // same-address atomics, ~45 ns each: 20 us of this 60 us kernel at T = 8192; pick_outliers_kernel folds the copies)
  unsigned *cm = colmax_bits + (size_t)(blockIdx.y % COLMAX_PARTS) * d;
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicMax(cm + c + e, __float_as_uint(m[e]));  // values >= 0
}

constexpr int MAX_OUT = 128;   // outlier dims fit one int8 k-tile
// single workgroup: dims whose column max exceeds 8x the mean column max (threshold raised until
// at most MAX_OUT qualify).  odims[0..MAX_OUT) = dim or -1, is_out[d] byte flags.
__global__ __launch_bounds__(1024) void pick_outliers_kernel(unsigned *__restrict__ colmax_bits, int d,
                                                             int *__restrict__ odims,
                                                             unsigned char *__restrict__ is_out) {
  __shared__ float red[16];
  __shared__ int s_cnt;
  for (int c = threadIdx.x; c < d; c += 1024) {          // fold the COLMAX_PARTS copies into the first (same thread reads it below)
    unsigned m = colmax_bits[c];
#pragma unroll
    for (int q = 1; q < COLMAX_PARTS; ++q) { const unsigned v = colmax_bits[(size_t)q * d + c]; m = v > m ? v : m; }
    colmax_bits[c] = m;
  }
  float sum = 0.f;
  for (int c = threadIdx.x; c < d; c += 1024) sum += __uint_as_float(colmax_bits[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = 0.f;
  for (int w = 0; w < 16; ++w) sum += red[w];
  float thr = 8.f * sum / d;
  for (int iter = 0; iter < 64; ++iter) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int c_loc = 0;
    for (int c = threadIdx.x; c < d; c += 1024) c_loc += (__uint_as_float(colmax_bits[c]) > thr) ? 1 : 0;
    if (c_loc) atomicAdd(&s_cnt, c_loc);
    __syncthreads();
    const int cnt = s_cnt;
    __syncthreads();
    if (cnt <= MAX_OUT) break;
    thr *= 1.5f;
  }
  for (int j = threadIdx.x; j < MAX_OUT; j += 1024) odims[j] = -1;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 1024) {
    const bool o = __uint_as_float(colmax_bits[c]) > thr;
    is_out[c] = o ? 1 : 0;
    if (o) odims[atomicAdd(&s_cnt, 1)] = c;   // order is irrelevant: A and B use the same list
  }
  __syncthreads();
  if (threadIdx.x == 0) odims[MAX_OUT] = s_cnt;   // the list is compact: the GEMM multiplies only ceil(count / 32) k-steps of the outlier tile
}

// one workgroup per token row (rows >= T of the padded tile are zero): per-token scales, int8 rows and
// the row constants of the error band, rowc[t] = (sx, m, P = z^2 |a_t|^2 / 12, 0)
// SRC = MSAE_F32 with x == a32 and b_dec == nullptr reads the prepared f32 activations; a shard of a feature-sharded
// group (nobody re-scores there) reads x - b_dec straight from the input instead and never writes a32.
template <int SRC, bool FROM_X>
__global__ __launch_bounds__(256) void quant_x_kernel(const void *__restrict__ x, const float *__restrict__ b_dec, int T, int d,
                                                      const int *__restrict__ odims,
                                                      const unsigned char *__restrict__ is_out,
                                                      signed char *__restrict__ xq,
                                                      signed char *__restrict__ xqo,
                                                      f32x4 *__restrict__ rowc, float zz12, int tile_major,
                                                      const unsigned *__restrict__ valid, unsigned need) {
  __shared__ float red[3][4];
  const int t = blockIdx.x;
  auto xq_at = [&](int c) { return xq + (tile_major ? packed_off((size_t)t, c, d, tile_major) : (size_t)t * d + c); };
  if (t >= T) {
    for (int c = threadIdx.x * 16; c < d; c += 4096) *reinterpret_cast<i32x4 *>(xq_at(c)) = i32x4{0, 0, 0, 0};
    if (threadIdx.x < 8) *reinterpret_cast<i32x4 *>(xqo + (size_t)t * MAX_OUT + threadIdx.x * 16) = i32x4{0, 0, 0, 0};
    if (threadIdx.x == 0) rowc[t] = f32x4{0.f, 1.f, 0.f, 0.f};
    return;
  }
  const float *__restrict__ row32 = static_cast<const float *>(x) + (size_t)t * d;   // SRC == MSAE_F32 && !FROM_X: a32
  auto load4 = [&](int c) {
    if constexpr (!FROM_X) {
      return *reinterpret_cast<const f32x4 *>(row32 + c);
    } else {
      f32x4 v = load_x4<SRC>(x, (size_t)t * d + c);
      if (b_dec) v = v - *reinterpret_cast<const f32x4 *>(b_dec + c);
      return v;
    }
  };
  float m_in = 0.f, m_out = 0.f, ss = 0.f;
  // a thread owns 16 consecutive dims of every 4096 (the 16 int8 it packs below); up to d = 8192 the row stays in registers
  // between the two passes
  constexpr int KEEP = 2;
  f32x4 keep[KEEP][4];
  unsigned keep_f[KEEP][4];
  const bool resident = d <= KEEP * 4096;
  {
    int it = 0;
    for (int c = threadIdx.x * 16; c < d; c += 4096, ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = load4(c + 4 * q);
        const unsigned flags = *reinterpret_cast<const unsigned *>(is_out + c + 4 * q);
        if (resident && it < KEEP) {
          if (it == 0) { keep[0][q] = v; keep_f[0][q] = flags; } else { keep[1][q] = v; keep_f[1][q] = flags; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = fabsf(v[e]);
          ss = __builtin_fmaf(av, av, ss);
          if ((flags >> (8 * e)) & 0xFFu) m_out = fmaxf(m_out, av); else m_in = fmaxf(m_in, av);
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m_in = fmaxf(m_in, __shfl_xor(m_in, off, 64));
    m_out = fmaxf(m_out, __shfl_xor(m_out, off, 64));
    ss += __shfl_xor(ss, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m_in; red[1][threadIdx.x >> 6] = m_out; red[2][threadIdx.x >> 6] = ss; }
  __syncthreads();
  m_in = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  m_out = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  ss = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  const float scale = m_in > 0.f ? m_in / 127.f : (m_out > 0.f ? m_out / 127.f : 1.f);
  int m = (int)ceilf(m_out / (127.f * scale));
  m = m < 1 ? 1 : (m > 32768 ? 32768 : m);   // the GEMM multiplies by m with a 24-bit multiply
  const float inv = 1.f / scale, inv_o = 1.f / (scale * (float)m);
  float e0 = 0.f;                              // energy of the non-outlier dims that round to zero (GUARD_E0_BANDS)
  int it2 = 0;
  for (int c = threadIdx.x * 16; c < d; c += 4096, ++it2) {
    i32x4 packed;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
      unsigned flags;
      if (resident) { v = it2 == 0 ? keep[0][q] : keep[1][q]; flags = it2 == 0 ? keep_f[0][q] : keep_f[1][q]; }
      else { v = load4(c + 4 * q); flags = *reinterpret_cast<const unsigned *>(is_out + c + 4 * q); }
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool outl = ((flags >> (8 * e)) & 0xFFu) != 0;
        int iv = outl ? 0 : (int)rintf(v[e] * inv);
        iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
        e0 += (!outl && iv == 0) ? v[e] * v[e] : 0.f;
        w |= ((unsigned)iv & 0xFFu) << (8 * e);
      }
      packed[q] = (int)w;
    }
    *reinterpret_cast<i32x4 *>(xq_at(c)) = packed;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) e0 += __shfl_xor(e0, off, 64);
  __syncthreads();                             // red[] of the first reduction has been consumed
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = e0;
  __syncthreads();
  if (threadIdx.x == 0) {
    e0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    // (stale operands, Prepared::valid: the candidate pass would read old weights -- every token to the exact path)
    const float guard = (e0 > GUARD_E0_SX * GUARD_E0_SX * scale * scale || (*valid & need) != need) ? 1.f : 0.f;
    rowc[t] = f32x4{scale, (float)m, zz12 * ss, guard};
  }
  if (threadIdx.x < MAX_OUT) {
    const int dim = odims[threadIdx.x];
    float av = 0.f;
    if (dim >= 0) {
      if constexpr (!FROM_X) av = row32[dim];
      else av = load_x1<SRC>(x, (size_t)t * d + dim) - (b_dec ? b_dec[dim] : 0.f);
    }
    int iv = dim >= 0 ? (int)rintf(av * inv_o) : 0;
    iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
    xqo[(size_t)t * MAX_OUT + threadIdx.x] = (signed char)iv;
  }
}

// bf16 pass: rowc[t] = (1, 1, P = z^2 * 5.5e-6 * |a_t|_4^2, 0); one 256-thread workgroup per token
__global__ __launch_bounds__(256) void row_p4_kernel(const float *__restrict__ a32, int T, int d,
                                                     f32x4 *__restrict__ rowc, float z2,
                                                     const unsigned *__restrict__ valid) {
  __shared__ float red[4];
  const int t = blockIdx.x;
  const float *row = a32 + (size_t)t * d;
  float s4 = 0.f;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float q = v[e] * v[e]; s4 = __builtin_fmaf(q, q, s4); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s4 += __shfl_xor(s4, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s4;
  __syncthreads();
  s4 = (red[0] + red[1]) + (red[2] + red[3]);
  if (threadIdx.x == 0) rowc[t] = f32x4{1.f, 1.f, z2 * BF16_REL_VAR2 * __builtin_sqrtf(s4), (*valid & PREP_BF16) ? 0.f : 1.f};
}

// Wq_o[n][j] = Wq[n][odims[j]] (0 where odims[j] < 0) for every feature row, and for the sample rows;
// with it the column constants of the error band for THIS batch's outlier dims:
//   colc[n] = (sw, Q, Si = |W_n|^2 - So, So = sum over outlier dims of (sw Wq)^2)   (colc_p: the same in main_row order)
__global__ __launch_bounds__(256) void gather_wo_kernel(const signed char *__restrict__ wq, int N, int d,
                                                        const int *__restrict__ odims,
                                                        const f32x4 *__restrict__ wstat,
                                                        signed char *__restrict__ wqo,
                                                        signed char *__restrict__ wqos,
                                                        f32x4 *__restrict__ colc, f32x4 *__restrict__ colc_s,
                                                        f32x4 *__restrict__ colc_p, int skip) {
  __shared__ int s_dims[MAX_OUT];
  if (threadIdx.x < MAX_OUT) s_dims[threadIdx.x] = odims[threadIdx.x];
  __syncthreads();
  const int n = blockIdx.x * 32 + (threadIdx.x >> 3);   // 8 threads per row, 16 bytes each (N % 32 == 0)
  const int j0 = (threadIdx.x & 7) * 16;
  // the tile is compact from column 0 and its readers stop after the k-steps that hold dims (32-B steps in gemm_mfma.h,
  // 64-B steps in gemm_skinny.h): columns from ceil(n_out / 64) * 64 on are neither gathered nor written
  const bool used = j0 < ((odims[MAX_OUT] + 63) & ~63);
  i32x4 packed = {0, 0, 0, 0};
  int sq = 0;
#pragma unroll
  for (int q = 0; q < 4 && used; ++q) {
    unsigned w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int dim = s_dims[j0 + 4 * q + e];
      const int v = dim >= 0 ? (int)wq[(size_t)n * d + dim] : 0;
      sq += v * v;
      w |= ((unsigned)v & 0xFFu) << (8 * e);
    }
    packed[q] = (int)w;
  }
  sq += __shfl_xor(sq, 1, 64);
  sq += __shfl_xor(sq, 2, 64);
  sq += __shfl_xor(sq, 4, 64);
  const bool samp = (n % SAMPLE_STRIDE) == SAMPLE_OFF;
  // skip: the main pass runs over the non-sample rows only (main_row); its outlier operand and column constants in that order
  if (used) {
    if (!skip) *reinterpret_cast<i32x4 *>(wqo + (size_t)n * MAX_OUT + j0) = packed;
    else if (!samp) *reinterpret_cast<i32x4 *>(wqo + (size_t)main_row(n) * MAX_OUT + j0) = packed;
    if (samp) *reinterpret_cast<i32x4 *>(wqos + (size_t)(n / SAMPLE_STRIDE) * MAX_OUT + j0) = packed;
  }
  if ((threadIdx.x & 7) == 0) {
    const f32x4 st = wstat[n];
    const float so = st[0] * st[0] * (float)sq;
    const f32x4 cc = {st[0], st[1], fmaxf(st[2] - so, 0.f), so};
    colc[n] = cc;
    if (samp) colc_s[n / SAMPLE_STRIDE] = cc;
    else if (skip) colc_p[main_row(n)] = cc;
  }
}

// The main pass leaves the sample features out (main_row): their candidates are the sample pass's own upper values above the
// token's threshold -- the entries the main pass's flush would have written for them (same u, same key).  One workgroup per token.
__global__ __launch_bounds__(256) void sample_push_kernel(const float *__restrict__ sample, int S,
                                                          const float *__restrict__ tau_vals, int tau_ld, int tau_col,
                                                          int skip_a, int skip_b, int *__restrict__ cnt,
                                                          unsigned long long *__restrict__ cand, int cap, int cnt_stride,
                                                          int row_stride) {
  const int t = blockIdx.x;
  const float tv = tau_vals[(size_t)t * tau_ld + tau_col];
  if (!(tv > 0.f)) return;                               // degenerate token: the main pass emits nothing either
  const float *row = sample + (size_t)t * S;
  for (int j = threadIdx.x * 4; j < S; j += 1024) {      // S % 256 == 0
    const f32x4 u = *reinterpret_cast<const f32x4 *>(row + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(u[e] > tv)) continue;
      const int feat = (j + e) * SAMPLE_STRIDE + SAMPLE_OFF;
      if (feat == skip_a || feat == skip_b) continue;
      const int slot = atomicAdd(cnt + (size_t)t * cnt_stride, 1);
      if (slot < cap) cand[(size_t)t * row_stride + slot] = ((unsigned long long)f32_order_key(u[e]) << 32) | (unsigned)(0x7FFFFFFF - feat);
    }
  }
}

// Segmented candidate lists (GemmEpilogue::segs, batches of few tokens) -> the contiguous list the consumers read.  One wave per
// token; a segment that overflowed reports cap + 1 (the consumers' "list overflow").
__global__ __launch_bounds__(64) void compact_candidates_kernel(const int *__restrict__ seg_cnt,
                                                                const unsigned long long *__restrict__ seg_cand, int segs,
                                                                int cap, int *__restrict__ cnt,
                                                                unsigned long long *__restrict__ cand) {
  const int t = blockIdx.x, lane = threadIdx.x, scap = cap / segs;
  int at = 0;
  bool over = false;
  for (int sg = 0; sg < segs; ++sg) {
    const int c = seg_cnt[(size_t)t * segs + sg];
    over |= c > scap;
    const int n = c < scap ? c : scap;
    for (int i = lane; i < n; i += 64) cand[(size_t)t * cap + at + i] = seg_cand[(size_t)t * cap + (size_t)sg * scap + i];
    at += n;
  }
  if (lane == 0) cnt[t] = over ? cap + 1 : at;
}

// Reference feature of the GEMM's separable band bound: refs = mean (Q, Si, So) over the sample rows'
// column constants (any positive triple is CORRECT -- the bound h_n B_t >= z sigma(t, n) holds by
// construction; a typical one makes it tight).  One workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void band_refs_kernel(const f32x4 *__restrict__ colc_s, int S, float *__restrict__ refs) {
  __shared__ float red[3][16];
  float q = 0.f, si = 0.f, so = 0.f;
  for (int j = threadIdx.x; j < S; j += 1024) {
    const f32x4 c = colc_s[j];
    q += c[1]; si += c[2]; so += c[3];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    q += __shfl_xor(q, off, 64); si += __shfl_xor(si, off, 64); so += __shfl_xor(so, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = q; red[1][threadIdx.x >> 6] = si; red[2][threadIdx.x >> 6] = so; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[threadIdx.x][w];
    refs[threadIdx.x] = fmaxf(t / (float)S, 1e-30f);
  }
}

// ---- MFMA GEMM: gemm_mfma.h.  Tile choice from tools/gemm_sweep on MI355X (T=8192, d=4096,
// N=131072): 256x256 tiles of 128-B k-rows, 2-slot ring, 8 waves as 2x4.
using GemmBf16 = GemmCfg<256, 256, 2, 2, 4, false>;
using GemmI8 = GemmCfg<256, 256, 2, 2, 4, true>;
#ifdef MSAE_GEMM_RING64
using GemmI8R64 = GemmCfg64<256, 256, 2, 4, true>;
#endif
constexpr int G_BM = GemmBf16::BM;

// ---- candidate select + exact re-score ----------------------------------------------------------
// The S = 1 weight stream reads every 1-KiB row piece exactly once per call: non-temporal loads (0.141 -> 0.131 ms
// at T = 1).  NOT for the re-scoring rows: a lane fetches a 128-B line in eight 16-B loads and lives on the
// cache holding it in between (non-temporal there: 1.18 -> 3.34 ms, profiles/r02_ab_nontemporal.txt).
#ifdef MSAE_GEMV_PLAIN_LOADS
#define MSAE_STREAM_LOAD(p) (*(p))
#else
#define MSAE_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifndef MSAE_RESCORE_U
#define MSAE_RESCORE_U 16
#endif
#ifndef MSAE_RESCORE_LPR      // lanes that share a row of W_enc in the FIRST round's re-scoring stream: 1, or 4 (64-B
#define MSAE_RESCORE_LPR 1    // pieces per row and instruction, 16 rows per pass: measured 1.61 ms against 1.17 -- not
#endif                        // the default).  Follow-up rounds of a few rows always use 4.
static_assert(MSAE_RESCORE_U * 4 == 64, "one re-scoring batch must be the 64 floats fast_shape_ok() guarantees");
struct RescoreArgs {
  const float *a32; const float *W_enc, *b_enc;
  const float *tau_vals; int tau_ld, tau_col;
  const int *cnt; const unsigned long long *cand; int cap;
  int T, d, N, k, r_max;
  int set_feature; float set_value; int zero_feature;
  const f32x4 *rowc, *colc;           // error-band constants per token / per feature
  float zz12, z2; int i8;
  float *vals; int32_t *idx; int64_t *idx64; int32_t *status;   // idx / idx64: either may be null
  int *flagged; int *n_flagged; int fb_cap;
  // EXT (feature-sharded group, msae_rescore_candidates): the candidate lists come as the shards' records
  // instead of cnt / cand / tau_vals / rowc / colc: record (g, t) at ext + ((size_t)g * ext_T + t) * ext_stride
  const unsigned char *ext; int ext_G, ext_C, ext_T, ext_stride, ext_valid;
  int lpr;   // lanes per row in the first round (1, 2, 4): small batches need the extra bytes in flight (rescore_shape)
};

// One shard's record of a token (msae_shard_candidates): C keys (order key of the upper value u | 0x7FFFFFFF -
// GLOBAL feature, 0 = empty), C times z sigma of that (token, feature) pair, tau = the largest u any feature of
// the shard NOT in the record can have (+inf: the shard could not bound it -> the token is recomputed exactly).
__host__ __device__ inline int shard_record_bytes(int C) { return C * 12 + 8; }

// Wave-wide bitonic sort (descending) of n = power-of-two u64 keys in LDS by ONE 64-lane wave.
template <int NT>
__device__ __forceinline__ void wave_sort_desc_u64(unsigned long long *s, int n, int lane) {
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = lane; i < (n >> 1); i += NT) {
        const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = s[lo], y = s[hi];
        if ((x < y) == desc) { s[lo] = y; s[hi] = x; }
      }
    }
  __syncthreads();
}

// number of keys (sorted descending, value in the upper 32 bits as an order key) whose value is >= v
__device__ __forceinline__ int count_ge(const unsigned long long *keys, int n, float v) {
  const unsigned tk = f32_order_key(v);
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((unsigned)(keys[mid] >> 32) >= tk) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ONE WAVE per token (64-thread workgroup; 4 waves for k > 64).  dynamic LDS: keys[cap] u64 | res[nrp] u64.
//
// The candidate list is ordered by the UPPER value u = coarse + z*sigma; lane c re-scores candidate c
// with the exact ascending-k f32 chain: it walks row f of W_enc with two software-pipelined batches
// of 16 x 16-B loads (256 B = two lines per batch) while the token's f32 activation vector a32[t][:]
// arrives through wave-uniform scalar loads.  No LDS staging of operands: the data in flight lives in
// VGPRs (7 waves x ~45 lanes x 512 B per CU), which is what keeps the HBM pipe full -- streaming the
// rows through LDS instead caps it at the ring size and measured 2.5 ms vs 1.4.
// HBM-bound: ~42 rows x d x 4 B per token.
//
// Rounds.  Needed are exactly the candidates with u >= v_k (the exact k-th value): everything else
// has p <= u < v_k.  v_k is not known beforehand, so round 1 takes the candidates with
//     u >= (k-th largest coarse value among the first NT) - zeta * (their median sigma)
// (the lanes look up the band of "their" candidate to get coarse = u - z*sigma), which is the needed
// set plus about one row in 96 % of the tokens; the exact v_k of round 1 is a lower bound of the final
// one, so ONE extension to every u >= v_k completes the rest.  A token verifies when
//     all candidates with u >= v_k are re-scored  and  v_k > tau  (non-candidates have u <= tau)
// and no re-scored pair contradicted the error model (|p - coarse| <= 6 sigma).  Tokens that fail (or
// overflowed their list / have tau <= 0 / more than r_max rows to read) go to the exact path.
// EXT: the list is the union of the shards' records; keys[] then carries the list POSITION in its low word
// (feature and z sigma are looked up by position: ef[], ezs[]).
#ifdef MSAE_RESCORE_TL   // tuning builds (tools/rescore_timeline.py): s_memtime stamps of thread 0 of the first 64 tokens
__device__ unsigned long long g_rs_tl[64 * 16];
#define MSAE_RTL(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (slot) < 16) g_rs_tl[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSAE_RTL(slot) do { } while (0)
#endif
// LDSA (small batches, p.lpr > 1): the token's activations are copied to LDS once and every lane reads the 16 B that
// belong to ITS piece of the row (ds_read_b128, counted waits) -- the scalar loads of the default path return out of
// order, so each pair of them is a full lgkmcnt(0) round trip (128 per pass), which nothing hides when a token's
// waves are alone on their SIMDs.
template <int NW, bool EXT = false, bool LDSA = false>   // NW waves per token: 1 for k <= 64, 4 for larger k (longer lists)
__global__ __launch_bounds__(64 * NW) void select_rescore_kernel(RescoreArgs p, const float *__restrict__ a32,
                                                            const float *__restrict__ W_enc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  const int nrp = next_pow2(p.r_max + 1);
  unsigned long long *res = keys + p.cap;
  [[maybe_unused]] float *ezs = reinterpret_cast<float *>(res + nrp);     // EXT only: [cap] z sigma by list position
  [[maybe_unused]] int *ef = reinterpret_cast<int *>(ezs + p.cap);        // EXT only: [cap] global feature by position
  [[maybe_unused]] float *a_lds = EXT ? reinterpret_cast<float *>(ef + p.cap) : ezs;   // LDSA only: [d]
  constexpr int NT = 64 * NW;
  __shared__ float s_cc[NT], s_zs[NT], s_pick[2];
  __shared__ int s_n;
  __shared__ unsigned s_tau;
  const int lane = threadIdx.x;   // thread index within the token's workgroup
  const int t = blockIdx.x;
  if constexpr (EXT) { if (t >= p.ext_valid) return; }
  int cnt, n;
  float tau;
  MSAE_RTL(0);
  const float *__restrict__ a = a32 + (size_t)t * p.d;  // noalias kernel arg + uniform address: s_load
  if constexpr (LDSA) {                                   // published by the barriers of the list sort below
    for (int i = 4 * (int)threadIdx.x; i < p.d; i += 4 * 64 * NW)
      *reinterpret_cast<f32x4 *>(a_lds + i) = *reinterpret_cast<const f32x4 *>(a + i);
  }
  f32x4 rc = {0.f, 0.f, 0.f, 0.f};
  const bool i8 = p.i8 != 0;
  int np;
  if constexpr (EXT) {
    const int M = p.ext_G * p.ext_C;
    np = next_pow2(M > 2 ? M : 2);
    if (lane == 0) { s_n = 0; s_tau = 0u; }
    __syncthreads();
    int mine = 0;
    for (int i = lane; i < np; i += NT) {
      unsigned long long kv = 0ull;
      if (i < M) {
        const int g = i / p.ext_C, j = i - g * p.ext_C;
        const unsigned char *rec = p.ext + ((size_t)g * p.ext_T + t) * p.ext_stride;
        const unsigned long long key = reinterpret_cast<const unsigned long long *>(rec)[j];
        if (key != 0ull) {
          kv = (key & 0xFFFFFFFF00000000ull) | (unsigned)(0x7FFFFFFF - i);
          ef[i] = rank_key_index(key);
          ezs[i] = reinterpret_cast<const float *>(rec + (size_t)p.ext_C * 8)[j];
          ++mine;
        }
      }
      keys[i] = kv;
    }
    if (mine) atomicAdd(&s_n, mine);
    for (int g = lane; g < p.ext_G; g += NT) {   // tau = the largest bound of ALL shards (order keys: +inf dominates, NaN never enters)
      const unsigned char *rec = p.ext + ((size_t)g * p.ext_T + t) * p.ext_stride;
      atomicMax(&s_tau, f32_order_key(*reinterpret_cast<const float *>(rec + (size_t)p.ext_C * 12)));
    }
    __syncthreads();
    n = cnt = s_n;
    tau = f32_from_order_key(s_tau);
  } else {
    cnt = p.cnt[t];
    n = cnt < p.cap ? cnt : p.cap;
    tau = p.tau_vals[(size_t)t * p.tau_ld + p.tau_col];
    rc = p.rowc[t];
    np = next_pow2(n > 2 ? n : 2);
  }
  for (int i = lane; i < nrp; i += NT) res[i] = 0ull;
  MSAE_RTL(1);
  // keys[0, n_sorted) hold the n_sorted largest keys in descending order (upper value desc, index asc on ties).
  // PARTIAL: of a list of ~650 candidates a token uses the first 40-60, so one wave first SELECTS its PRE_LO..PRE_HI
  // largest (keys in registers, bisection on the value word with ballot counts, a handful of steps) and sorts only
  // those 128 slots; whoever then needs a candidate behind them (count_needed, the target check of a round) gets the
  // full sort after all -- the presorted prefix is the same keys in the same places.
  constexpr int PRE_LO = 96, PRE_HI = 128, PRE_MIN = 192, PRE_PK = 32;
  int n_sorted = n;
  bool partial = false;
  auto full_sort = [&]() {
    if constexpr (!EXT) {
      __syncthreads();
      for (int i = lane; i < np; i += NT) keys[i] = (i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
    }
    wave_sort_desc_u64<NT>(keys, np, lane);
  };
  if constexpr (!EXT && NW == 1) {
#ifndef MSAE_RESCORE_NO_PRESELECT
    if (n > PRE_MIN && n <= 64 * PRE_PK && p.k + 4 <= 64) {          // wave-uniform
      const int nj = (n + 63) >> 6;
      unsigned long long kreg[PRE_PK];
#pragma unroll
      for (int j = 0; j < PRE_PK; ++j) {
        const int i = j * 64 + lane;
        kreg[j] = (j < nj && i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
      }
      unsigned lo = 0u, hi = 0xFFFFFFFFu;     // count(value word >= lo) > PRE_HI, count(>= hi) < PRE_LO
      int c_sel = -1;
      unsigned thr = 0u;
      while (hi - lo > 1u) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int jb = 0; jb < PRE_PK; jb += 8) {             // one branch per eight key slots (empty slots hold 0)
          if (jb < nj) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j)
              c += __builtin_popcountll(__builtin_amdgcn_ballot_w64((unsigned)(kreg[j] >> 32) >= mid));
          }
        }
        if (c > PRE_HI) lo = mid;
        else if (c < PRE_LO) hi = mid;
        else { c_sel = c; thr = mid; break; }
      }
      if (c_sel > 0) {                                       // (ties across the window: no such threshold -> full sort)
        int base = 0;
#pragma unroll
        for (int j = 0; j < PRE_PK; ++j) {
          if (j < nj) {
            const bool take = (unsigned)(kreg[j] >> 32) >= thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
            if (take) keys[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = kreg[j];
            base += __builtin_popcountll(m);
          }
        }
        for (int i = c_sel + lane; i < PRE_HI; i += NT) keys[i] = 0ull;
        wave_sort_desc_u64<NT>(keys, PRE_HI, lane);
        partial = true;
        n_sorted = c_sel;
      }
    }
#endif
  }
  if (!partial) full_sort();
  auto need_full = [&]() { full_sort(); partial = false; n_sorted = n; };
  auto count_needed = [&](float v) {         // candidates with u >= v (over the whole list)
    int c = count_ge(keys, n_sorted, v);
    if (partial && c >= n_sorted) { need_full(); c = count_ge(keys, n, v); }
    return c;
  };
  MSAE_RTL(2);
  const int has_set = p.set_feature >= 0 ? 1 : 0;
  if (lane == 0 && has_set) res[0] = rank_key(p.set_value, p.set_feature);

  // ---- size of the first round ------------------------------------------------------------------
  const int lim = n < p.r_max ? n : p.r_max;
  int target = lim;
  {
    const int mt_max = p.k <= 64 ? 64 : NT;       // the same statistic whatever the number of waves per token
    const int mt = n < mt_max ? n : mt_max;
    float my_cc = -__builtin_inff(), my_zs = 0.f;
    if (lane < mt) {
      const unsigned long long key = keys[lane];
      if constexpr (EXT) my_zs = ezs[rank_key_index(key)];
      else my_zs = __builtin_sqrtf(band_sq(rc, p.colc[rank_key_index(key)], p.zz12, i8));
      my_cc = f32_from_order_key((unsigned)(key >> 32)) - my_zs;
    }
    s_cc[lane] = my_cc;
    s_zs[lane] = my_zs;
    if (lane < 2) s_pick[lane] = lane == 0 ? -__builtin_inff() : 0.f;
    __syncthreads();
    const int kk = p.k - has_set;
    if (lane < mt && kk >= 1 && kk <= mt) {
      int rank_c = 0, rank_z = 0;
      for (int j = 0; j < mt; ++j) {
        const float cj = s_cc[j], zj = s_zs[j];
        rank_c += (cj > my_cc || (cj == my_cc && j < lane)) ? 1 : 0;
        rank_z += (zj < my_zs || (zj == my_zs && j < lane)) ? 1 : 0;
      }
      if (rank_c == kk - 1) s_pick[0] = my_cc;
      if (rank_z == mt / 2) s_pick[1] = my_zs;
    }
    __syncthreads();
    if (kk >= 1 && kk <= mt && p.z2 > 0.f) {
      const float thr1 = s_pick[0] - GUARD_ZETA * s_pick[1] * __builtin_amdgcn_rsqf(p.z2);
      int n1 = count_needed(thr1);
      if (n1 < p.k + 4) n1 = p.k + 4;
      target = n1 < lim ? n1 : lim;
    }
  }
  if constexpr (LDSA) {   // small batch: one pass reads 64 NW / lpr rows whatever the target -- fill it (fewer second rounds)
    const int rpp = p.lpr > 0 ? NT / p.lpr : NT;
    const int fill = rpp < lim ? rpp : lim;
    if (target < fill) target = fill;
  }

  MSAE_RTL(3);
  const float zc2 = GUARD_Z_CHECK * GUARD_Z_CHECK;
  const bool guarded = !EXT && rc[3] != 0.f;     // the token's shape is outside the noise model (quant_x_kernel): exact path
  if (guarded) target = 0;                       // (no row is read for it here)
  int done = 0;                                  // candidates re-scored so far (wave-uniform)
  bool ok = false, viol = false;
  int rounds = 0;
  const int first_target = target;
  (void)first_target; (void)rounds;
  for (;;) {
    ++rounds;
    int my_viol = 0;
    // LPR = 1: lane c streams row c (16 B per lane and instruction).  LPR = 4 (tuning builds): four lanes share a
    // row, lane q loading bytes [16 q, 16 q + 16) of every 64-B piece -- four times fewer cache lines per
    // instruction, but only 16 rows per pass, i.e. three row-streaming latencies per round instead of one.  The
    // chain stays one serial ascending-k sequence: sub-step q multiplies the group's lane-q piece (every lane
    // executes it on its own registers; only lane q's is the true partial sum) and a quad rotate hands the
    // accumulator on.  The activations are wave-uniform scalar operands either way.
    auto run_pass = [&](auto lpr_tag) {
      constexpr int LPR = decltype(lpr_tag)::value;
      constexpr int RPP = NT / LPR;                  // rows per pass
      constexpr int RS_U = MSAE_RESCORE_U, RS_B = 4 * RS_U * LPR;   // floats of a row per batch
      const int rq = lane / LPR, q = lane % LPR;
      for (int c0 = done; c0 < target; c0 += RPP) {
        const int c = c0 + rq;
        const bool active = c < target;
        const unsigned long long key = active ? keys[c] : keys[c0];
        int f = rank_key_index(key);
        float ext_zs = 0.f;
        f32x4 cc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EXT) { ext_zs = ezs[f]; f = ef[f]; }     // list position -> (z sigma, global feature)
        else cc = p.colc[f];
        const float upper = f32_from_order_key((unsigned)(key >> 32));
        const float *__restrict__ w = W_enc + (size_t)f * p.d + 4 * q;
        float acc = 0.f;
        // two batches of RS_U x 16 B per lane, software-pipelined: while one batch is consumed the
        // other is in flight, so the lane never drains its loads
        f32x4 wa[RS_U], wb[RS_U];
        auto fetch = [&](f32x4 (&dst)[RS_U], int kk) {
#pragma unroll
          for (int u = 0; u < RS_U; ++u) dst[u] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * LPR * u);
        };
        auto consume = [&](const f32x4 (&src)[RS_U], int kk) {
#pragma unroll
          for (int u = 0; u < RS_U; ++u) {
            [[maybe_unused]] f32x4 av;                   // LDSA: the activations of this lane's own piece
            if constexpr (LDSA) av = *reinterpret_cast<const f32x4 *>(a_lds + kk + 4 * LPR * u + 4 * q);
#pragma unroll
            for (int qq = 0; qq < LPR; ++qq) {
              const int k0 = kk + 4 * LPR * u + 4 * qq;
              if constexpr (LDSA) {                      // only sub-step qq == q carries the true partial sum
                acc = __builtin_fmaf(av[0], src[u][0], acc);
                acc = __builtin_fmaf(av[1], src[u][1], acc);
                acc = __builtin_fmaf(av[2], src[u][2], acc);
                acc = __builtin_fmaf(av[3], src[u][3], acc);
              } else {
                acc = __builtin_fmaf(a[k0 + 0], src[u][0], acc);   // a[] is wave-uniform: SGPRs
                acc = __builtin_fmaf(a[k0 + 1], src[u][1], acc);
                acc = __builtin_fmaf(a[k0 + 2], src[u][2], acc);
                acc = __builtin_fmaf(a[k0 + 3], src[u][3], acc);
              }
              if constexpr (LPR == 4)   // quad_perm:[3,0,1,2] -- lane i takes lane i - 1's value, lane 0 lane 3's
                acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x93, 0xF, 0xF, false));
              if constexpr (LPR == 2)   // quad_perm:[1,0,3,2] -- the two lanes of a pair swap
                acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, false));
            }
          }
        };
        fetch(wa, 0);
        for (int kk = 0; kk < p.d; kk += 2 * RS_B) {     // d % RS_B == 0 (fast_shape_ok / the caller's choice of LPR)
          const bool has_b = kk + RS_B < p.d;
          if (has_b) fetch(wb, kk + RS_B);
          consume(wa, kk);
          if (kk + 2 * RS_B < p.d) fetch(wa, kk + 2 * RS_B);
          if (has_b) consume(wb, kk + RS_B);
        }
        const float pre = acc + (p.b_enc ? p.b_enc[f] : 0.f);
        if (active && q == 0) {                          // whole pieces done: the sum is back in the group's lane 0
          res[has_set + c] = rank_key(pre > 0.f ? pre : 0.f, f);  // slots past the sorted prefix are 0
          // model check: |p - coarse| <= 6 sigma  <=>  (p - coarse)^2 z^2 <= 36 (z sigma)^2
          const float zs2 = EXT ? ext_zs * ext_zs : band_sq(rc, cc, p.zz12, i8);
          const float diff = pre - (upper - __builtin_sqrtf(zs2));
          if (diff * diff * p.z2 > zc2 * zs2 * 1.0001f + 1e-30f) my_viol = 1;
        }
      }
    };
    // A follow-up round re-scores a handful of rows: with a lane per row each of them is a latency chain (16 KB at
    // 512 B in flight = 32 round trips, ~60 us whatever the load); four lanes per row carry 2 KB in flight each.
    // The first round of a SMALL batch (too few tokens to fill the chip with a lane per row) does the same with
    // p.lpr lanes per row and as many waves per token.
    if (partial && target > n_sorted) need_full();          // wave-uniform
    const bool few = rounds > 1 && target - done <= NT / 4;
    int lpr = few ? 4 : (MSAE_RESCORE_LPR == 4 ? 4 : p.lpr);
    while (lpr > 1 && p.d % (4 * MSAE_RESCORE_U * lpr) != 0) lpr >>= 1;      // a batch is 64 lpr floats of a row
    if (lpr == 4) run_pass(std::integral_constant<int, 4>());
    else if (lpr == 2) run_pass(std::integral_constant<int, 2>());
    else run_pass(std::integral_constant<int, 1>());
    done = target;
    viol = viol || (__syncthreads_or(my_viol) != 0);
    MSAE_RTL(2 + 2 * rounds);
    {   // res[] is zero (= empty, the smallest key) behind the slots written so far: sort the filled prefix only
      const int filled = next_pow2(done + has_set > 2 ? done + has_set : 2);
      wave_sort_desc_u64<NT>(res, filled < nrp ? filled : nrp, lane);
    }
    MSAE_RTL(3 + 2 * rounds);
    const bool have_k = done + has_set >= p.k;
    const float v_k = f32_from_order_key((unsigned)(res[p.k - 1] >> 32));
    const int needed = have_k ? count_needed(v_k) : n;          // candidates with u >= v_k
    ok = (cnt <= p.cap) && (tau > 0.f) && have_k && !viol && needed <= done && v_k > tau * 1.000001f && !guarded;
    if (ok || viol || guarded || done >= lim || !(tau > 0.f) || cnt > p.cap) break;
    target = needed > done ? needed : done + 1;
    if (target > lim) target = lim;
    __syncthreads();
  }

  MSAE_RTL(14);
#ifdef MSAE_RESCORE_TL
  if (threadIdx.x == 0 && blockIdx.x < 64) g_rs_tl[blockIdx.x * 16 + 15] = ((unsigned long long)rounds << 32) | (unsigned)done;
#endif
  for (int j = lane; j < p.k; j += NT) {
    const unsigned long long key = res[j];
    const int fi = key ? rank_key_index(key) : 0;
    if (p.idx) p.idx[(size_t)t * p.k + j] = fi;
    if (p.idx64) p.idx64[(size_t)t * p.k + j] = fi;
    p.vals[(size_t)t * p.k + j] = key ? f32_from_order_key((unsigned)(key >> 32)) : 0.f;
  }
  if (lane == 0) {
    // not verified: 2 | reason bits (4 list overflow, 8 tau <= 0, 16 fewer than k candidates,
    // 32 more than r_max rows needed / v_k not above tau, 64 a re-scored pair contradicted the error
    // model); the exact fallback rewrites it to 1 once it has recomputed t
    const int reason = 2 | (cnt > p.cap ? 4 : 0) | (!(tau > 0.f) ? 8 : 0) |
                       (done + has_set < p.k ? 16 : 0) | (guarded ? 128 : (viol ? 64 : 32));
    if (p.status) p.status[t] = ok ? 0 : reason;
#ifdef MSAE_RESCORE_DEBUG   // rows / rounds histogram (tools/rescore_stats.py); breaks the status contract
    if (p.status && ok) p.status[t] = (rounds << 24) | (first_target << 12) | done;
#endif
    if (!ok) {
      const int slot = atomicAdd(p.n_flagged, 1);
      if (slot < p.fb_cap) p.flagged[slot] = t;
    }
  }
}

// Feature-sharded group, sender side: the C best candidates of THIS shard per token by upper value, as the
// record shard_record_bytes() describes (global feature ids).  One wave per token.
struct PackArgs {
  const int *cnt; const unsigned long long *cand; int cap;
  const float *tau_vals; int tau_ld, tau_col;
  const f32x4 *rowc, *colc; float zz12; int i8;
  int C, row_offset, stride;
  unsigned char *recs;
};
template <int PK>   // key slots per lane: the list (<= cap <= 64 PK keys) lives in registers
__global__ __launch_bounds__(64) void pack_candidates_kernel(PackArgs p) {
  // The C largest of ~512 keys are a selection, not a sort: the keys sit in registers (PK per lane) and a bisection
  // on the 64-bit key -- unique: the feature id is its low word -- finds the C-th largest with one ballot count
  // per key slot and step; the survivors are compacted with ballot prefix counts (any order: the owner sorts).
  const int t = blockIdx.x, lane = threadIdx.x;
  const int cnt = p.cnt[t];
  const int n = cnt < p.cap ? cnt : p.cap;
  const float tau = p.tau_vals[(size_t)t * p.tau_ld + p.tau_col];
  // list complete, a real threshold behind it, and a token the noise model describes (rowc[3]: quant_x_kernel's guard)
  const bool bounded = cnt <= p.cap && tau > 0.f && p.rowc[t][3] == 0.f;
  const int nj = bounded ? (n + 63) >> 6 : 0;            // key slots in use (wave-uniform)
  unsigned long long kreg[PK];
#pragma unroll
  for (int j = 0; j < PK; ++j) {
    const int i = j * 64 + lane;
    kreg[j] = (j < nj && i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
  }
  unsigned long long lo = 0ull, hi = ~0ull;              // count(key >= lo) >= C  (or everything is taken), count(>= hi) < C
  if (n > p.C) {
    while (hi - lo > 1ull) {
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int jb = 0; jb < PK; jb += 8) {               // one branch per eight key slots (empty slots hold 0 < mid)
        if (jb < nj) {
#pragma unroll
          for (int j = jb; j < jb + 8; ++j) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kreg[j] >= mid));
        }
      }
      if (c >= p.C) lo = mid; else hi = mid;
    }
  } else {
    lo = 1ull;                                           // every (non-empty) key
  }
  unsigned char *rec = p.recs + (size_t)t * p.stride;
  unsigned long long *okeys = reinterpret_cast<unsigned long long *>(rec);
  float *ozs = reinterpret_cast<float *>(rec + (size_t)p.C * 8);
  const f32x4 rc = p.rowc[t];
  int base = 0;
  unsigned long long below = 0ull;                       // largest key NOT taken
#pragma unroll
  for (int j = 0; j < PK; ++j) {
    if (j < nj) {
      const unsigned long long key = kreg[j];
      const bool take = key >= lo && key != 0ull;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
      if (take) {
        const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        const int f = rank_key_index(key);
        okeys[pos] = (key & 0xFFFFFFFF00000000ull) | (unsigned)(0x7FFFFFFF - (f + p.row_offset));
        ozs[pos] = __builtin_sqrtf(band_sq(rc, p.colc[f], p.zz12, p.i8 != 0));
      } else {
        below = key > below ? key : below;
      }
      base += __builtin_popcountll(m);
    }
  }
  for (int jj = base + lane; jj < p.C; jj += 64) { okeys[jj] = 0ull; ozs[jj] = 0.f; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(below, off, 64);
    below = o > below ? o : below;
  }
  if (lane == 0) {
    // what the shard's other features can reach: the best candidate left behind, else the threshold every
    // non-candidate stayed below; +inf when the shard cannot tell (overflowed list, degenerate token)
    float b = __builtin_inff();
    if (bounded) b = below != 0ull ? f32_from_order_key((unsigned)(below >> 32)) : tau;
    float *tail = reinterpret_cast<float *>(rec + (size_t)p.C * 12);
    tail[0] = b;
    tail[1] = 0.f;
  }
}

// waves per token and lanes per row of the first round: k > 64 -> 4 waves (longer lists); batches that cannot fill
// 256 CUs x 8 waves with a lane per row get 2 or 4 lanes per row (and waves per token) instead
inline void rescore_shape(int T, int k, int &nw, int &lpr) {
  // k > 64: 4 waves per token, a lane per row.  k = 256 reads ~350 rows per token (profiles/r03_rescore_stats_k256.txt),
  // i.e. a second, mostly idle pass -- but 6 waves per token (one pass) measured SLOWER, 9.65 vs 8.0 ms: the kernel's
  // ~230 VGPRs allow 8 waves per CU, and workgroups of 6 waves leave two of those slots empty
  // (profiles/r03_k256_nw6.txt); the stage is HBM-bound at 5.8 TB/s either way.
  nw = k <= 64 ? 1 : 4;
  lpr = 1;
  if (k <= 64) {
    const long lanes = (long)T * (k + 13);
    if (lanes * 4 <= 131072) lpr = 4;
    else if (lanes * 2 <= 131072) lpr = 2;
    nw = lpr;
  }
}
template <bool EXT>
inline int launch_select_rescore(RescoreArgs &ra, int T, int k, size_t smem, const float *a32, const float *W_enc,
                                 hipStream_t s) {
  int nw;
  rescore_shape(T, k, nw, ra.lpr);
  const bool ldsa = ra.lpr > 1 && smem + (size_t)ra.d * 4 <= 96 * 1024;      // small batch: activations in LDS
  if (ldsa) smem += (size_t)ra.d * 4;
#define MSAE_RS_LAUNCH(NWV, LDSAV)                                                                                   \
  do {                                                                                                               \
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)select_rescore_kernel<NWV, EXT, LDSAV>,                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                        \
    hipLaunchKernelGGL((select_rescore_kernel<NWV, EXT, LDSAV>), dim3(T), dim3(64 * NWV), smem, s, ra, a32, W_enc);  \
  } while (0)
  if (ldsa) { if (nw == 2) MSAE_RS_LAUNCH(2, true); else MSAE_RS_LAUNCH(4, true); }
  else if (nw == 1) MSAE_RS_LAUNCH(1, false);
  else if (nw == 2) MSAE_RS_LAUNCH(2, false);
  else MSAE_RS_LAUNCH(4, false);
#undef MSAE_RS_LAUNCH
  return 0;
}

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

// index output of one call: 32-bit (msae_encode_topk), 64-bit (msae_encode_topk_i64), never both null
struct IdxOut { int32_t *i32; int64_t *i64; };
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
// ---- small-T path (steering decode steps, S = 1: features/steering.py:86,105-124) -------------------------
// T <= 4 tokens cannot feed a 256-row MFMA tile; the pass is a 0.5 GiB weight stream, so it is written as
// one: every wave walks rows of Wq with 16-B lane loads and v_dot4_i32_i8 against the tokens' activations
// held in registers.  The activations are quantised to 15 bits as TWO int8 planes (a ~ s (128 hi + lo)),
// which removes the massive-activation problem without the per-batch outlier machinery (column maxima,
// outlier tile of Wq): the x-side rounding noise becomes negligible and the band constants are static.
//   prep_small    a32, two-plane quantisation, rowc = (s, 1, P = z^2 |a|^2 / 12)
//   gemv_small    u = coarse + z sigma of every row; each workgroup keeps the upper values of ITS rows (<= 128)
//                 in LDS and emits its SMALL_EMIT best as (u, feature) keys plus its next value as a bound
//   select_small  one workgroup per token: a threshold (bisection on the value) with SMALL_R .. SMALL_RMAX of the
//                 SMALL_GRID x SMALL_EMIT survivors at or above it; those are the candidates, and
//                 tau = max(survivors below it, every workgroup's bound) bounds all other features
//   rescore_small one WAVE per (token, candidate): row and activations in registers (lane l holds elements
//                 256 c + 4 l ..), the exact ascending-k chain walks the lanes (4 fma + a one-lane wave
//                 rotate per step, ~6 cycles per element instead of ~15 for a one-lane chain out of LDS);
//                 the LAST wave of a token to finish sorts the exact values and writes the outputs: verified
//                 iff v_k lies above tau
constexpr int SMALL_T_MAX = 16, SMALL_T_DOT4 = 4;      // small path: T <= 16 (d <= 4096), T <= 4 for wider inputs
constexpr int SMALL_DOT4_PREF = 1;                     // T = 1: dot4 stream (0.130 ms vs 0.154); T >= 2: MFMA stream
                                                       // (T = 2 / 3 / 4: 0.158 / 0.156 / 0.159 ms vs 0.165 / 0.205 / 0.213)
constexpr int SMALL_R = 96, SMALL_RMAX = 127, SMALL_K_MAX = 64;   // candidates per token: R .. RMAX
constexpr int SMALL_MF_EMIT = 5;                       // MFMA stream: survivors per workgroup (one per CU) and token
constexpr int SMALL_GRID = 2048;                       // gemv workgroups of 4 waves (8 per CU)
constexpr int SMALL_EMIT = 3;                          // survivors per workgroup and token
constexpr int SMALL_WG_ROWS = 128;                     // most rows of one workgroup (32 per wave)
constexpr int SMALL_SURV = SMALL_GRID * SMALL_EMIT;    // 6144 keys per token
static_assert(SMALL_RMAX < SMALL_SURV && SMALL_RMAX + 1 <= 128, "candidate list: 127 exact values + the hook's set_feature");
inline bool small_shape_ok(int T, int d, int N, int k) {
  return T <= SMALL_T_MAX && (T <= SMALL_T_DOT4 || d <= 4096) && k <= SMALL_K_MAX && d % 1024 == 0 && d <= 8192 &&
         N >= 4096 && N <= SMALL_GRID * SMALL_WG_ROWS && i8_shape_ok(N, d);
}

struct FusedPlan {
  bool fast, i8, small;
  size_t off_xhi, off_xlo, off_skeys, off_sviol, off_surv, off_sbound, off_scand, off_stau;
  int Tp, S, r, cap, r_max, fb_cap, fb_chunks;
  size_t off_xq, off_xqo, off_rowc, off_refs, off_colc, off_colc_s, off_colc_p, off_colmax, off_odims, off_isout, off_wqo, off_wqos;
  size_t off_xb, off_a32, off_sample, off_tauv, off_taui, off_cnt, off_cand, off_segcnt, off_segcand, off_flag, off_fbdense, off_dense, bytes;
  int segs;   // > 1: the candidate passes append to segmented lists (compact_candidates_kernel joins them)
};

inline FusedPlan make_plan(int T, int d, int N, int k, int mode, int shard_C = 0) {
  FusedPlan p{};
  p.fast = fast_shape_ok(N, d) && T > EXACT_T_MAX && k <= 256 && k >= 1;
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
    p.small = p.i8 && small_shape_ok(T, d, N, k) && getenv("MSAE_NO_SMALL_PATH") == nullptr;
    // most rows one token may read before it is handed to the exact path (the needed set is ~k + 10:
    // reaching this means the band is not separating anything); at least k + 4 (first-round minimum)
    p.r_max = k <= 64 ? 8 * k : 3 * k;
    if (p.r_max < k + 4) p.r_max = k + 4;
    if (p.r_max > p.cap) p.r_max = p.cap;
    if (p.i8) {
      p.off_xq = take((size_t)p.Tp * d);
      p.off_xqo = take((size_t)p.Tp * MAX_OUT);
      p.off_colc = take((size_t)N * 16);
      p.off_colc_s = take((size_t)p.S * 16);
      p.off_colc_p = take((size_t)N * 16);
      p.off_colmax = take((size_t)d * 4 * COLMAX_PARTS);
      p.off_odims = take((size_t)(MAX_OUT + 1) * 4);
      p.off_isout = take((size_t)d);
      p.off_wqo = take((size_t)N * MAX_OUT);
      p.off_wqos = take((size_t)p.S * MAX_OUT);
    }
    p.off_rowc = take((size_t)p.Tp * 16);
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
    p.off_xb = take(p.i8 ? 256 : (size_t)p.Tp * d * 2);
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

// ---- small-T path kernels ---------------------------------------------------------------------------------
// one 256-thread workgroup per token: a32, two-plane quantisation q = rint(a / s), q = 128 hi + lo with
// hi in [-127, 127], lo in [-64, 63], s = max|a| / 16319; rowc[t] = (s, 1, z^2 |a|^2 / 12, 0)
template <int DT>
__global__ __launch_bounds__(256) void prep_small_kernel(const void *__restrict__ x, const float *__restrict__ b_dec,
                                                         int d, float *__restrict__ a32, signed char *__restrict__ xhi,
                                                         signed char *__restrict__ xlo, f32x4 *__restrict__ rowc,
                                                         float zz12, int *__restrict__ zero_a, int n_a,
                                                         int *__restrict__ zero_b, int n_b,
                                                         const unsigned *__restrict__ valid, unsigned need, int T) {
  __shared__ float red[2][4];
  const int t = blockIdx.x;
  if (t == 0) {   // per-call counters (model-check flags [T], finished-wave counters [T], flag list + counts) start at zero
    // stale operands (Prepared::valid): the model-check flag of every token starts RAISED -- all of them go to the exact path
    const int stale = (*valid & need) != need ? 1 : 0;
    for (int i = threadIdx.x; i < n_a; i += 256) zero_a[i] = i < T ? stale : 0;
    for (int i = threadIdx.x; i < n_b; i += 256) zero_b[i] = 0;
  }
  float m = 0.f, ss = 0.f;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    f32x4 v = load_x4<DT>(x, (size_t)t * d + c);
    if (b_dec) v = v - *reinterpret_cast<const f32x4 *>(b_dec + c);
    *reinterpret_cast<f32x4 *>(a32 + (size_t)t * d + c) = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { m = fmaxf(m, fabsf(v[e])); ss = __builtin_fmaf(v[e], v[e], ss); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { m = fmaxf(m, __shfl_xor(m, off, 64)); ss += __shfl_xor(ss, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float scale = m > 0.f ? m / 16319.f : 1.f;
  if (threadIdx.x == 0) rowc[t] = f32x4{scale, 1.f, zz12 * ss, 0.f};
  const float inv = 1.f / scale;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(a32 + (size_t)t * d + c);
    unsigned wh = 0, wl = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q = (int)rintf(v[e] * inv);
      q = q > 16319 ? 16319 : (q < -16319 ? -16319 : q);
      const int hi = (q + 64) >> 7, lo = q - hi * 128;
      wh |= ((unsigned)hi & 0xFFu) << (8 * e);
      wl |= ((unsigned)lo & 0xFFu) << (8 * e);
    }
    *reinterpret_cast<unsigned *>(xhi + (size_t)t * d + c) = wh;
    *reinterpret_cast<unsigned *>(xlo + (size_t)t * d + c) = wl;
  }
}

// The weight stream.  A wave owns rows n = w, w + W, ... (W waves): per row DSEG loads of 16 B per lane (1 KiB per
// instruction), 8 dot4 per segment and token, a wave reduction, u = coarse + z sigma.  HBM-bound: N d bytes
// once, whatever T <= 4.  The workgroup's upper values go to LDS as rank keys; at the end wave t picks
// token t's SMALL_EMIT + 1 largest (four max-reductions) -> surv[t][wg][0..EMIT), bound[t][wg].
template <int DSEG, int TT>
__global__ __launch_bounds__(256) void gemv_small_kernel(const signed char *__restrict__ wq, const f32x4 *__restrict__ wstat,
                                                         const float *__restrict__ b_enc, int N, int T,
                                                         const signed char *__restrict__ xhi,
                                                         const signed char *__restrict__ xlo,
                                                         const f32x4 *__restrict__ rowc, float zz12, int skip_a,
                                                         int skip_b, unsigned long long *__restrict__ surv,
                                                         unsigned *__restrict__ bound) {
  constexpr int d = DSEG * 1024;
  __shared__ unsigned long long wgk[TT][SMALL_WG_ROWS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  for (int i = threadIdx.x; i < TT * SMALL_WG_ROWS; i += 256) (&wgk[0][0])[i] = 0ull;
  i32x4 xh[TT][DSEG], xl[TT][DSEG];
  float sxz[TT], pz[TT], rz[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tt = t < T ? t : T - 1;
#pragma unroll
    for (int q = 0; q < DSEG; ++q) {
      xh[t][q] = *reinterpret_cast<const i32x4 *>(xhi + (size_t)tt * d + q * 1024 + lane * 16);
      xl[t][q] = *reinterpret_cast<const i32x4 *>(xlo + (size_t)tt * d + q * 1024 + lane * 16);
    }
    const f32x4 rc = rowc[tt];
    sxz[t] = rc[0]; pz[t] = rc[2]; rz[t] = rc[0] * rc[0] * zz12;
  }
  __syncthreads();
  constexpr int RB = 16 / DSEG > 0 ? 16 / DSEG : 1;      // rows in flight per wave: 16 KiB of loads outstanding
  int slot = wv * (SMALL_WG_ROWS / 4);                     // this wave's next key slot (<= 32 rows per wave)
  // rows wave, wave + W, wave + 2 W, ...: every workgroup sees a thin, index-strided slice of the features
  for (int n0 = wave; n0 < N; n0 += n_waves * RB) {
    i32x4 w[RB][DSEG];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int q = 0; q < DSEG; ++q)
        w[r][q] = (n0 + r * n_waves < N)
                      ? MSAE_STREAM_LOAD(reinterpret_cast<const i32x4 *>(wq + (size_t)(n0 + r * n_waves) * d + q * 1024 + lane * 16))
                      : i32x4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int n = n0 + r * n_waves;
      const bool live = n < N;                          // wave-uniform
      const f32x4 st = wstat[live ? n : 0];
      const float bias = b_enc ? b_enc[live ? n : 0] : 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        int ah = 0, al = 0;
#pragma unroll
        for (int q = 0; q < DSEG; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ah = __builtin_amdgcn_sdot4(w[r][q][e], xh[t][q][e], ah, false);
            al = __builtin_amdgcn_sdot4(w[r][q][e], xl[t][q][e], al, false);
          }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ah += __shfl_xor(ah, off, 64); al += __shfl_xor(al, off, 64); }
        if (lane == 0 && t < T && live) {
          const float c = (128.f * (float)ah + (float)al) * (sxz[t] * st[0]) + bias;
          const float zs = __builtin_sqrtf(__builtin_fmaf(pz[t], st[1], rz[t] * st[2]));
          wgk[t][slot + r] = rank_key((n == skip_a || n == skip_b) ? -__builtin_inff() : c + zs, n);
        }
      }
    }
    slot += RB;
  }
  __syncthreads();
  if (wv >= TT || wv >= T) return;
  unsigned long long k0 = wgk[wv][lane], k1 = wgk[wv][lane + 64], best[SMALL_EMIT + 1];
#pragma unroll
  for (int e = 0; e <= SMALL_EMIT; ++e) {
    unsigned long long m = k0 > k1 ? k0 : k1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off, 64);
      m = o > m ? o : m;
    }
    best[e] = m;                              // keys are unique (feature in the low word) unless 0 = empty
    if (k0 == m) k0 = 0ull; else if (k1 == m) k1 = 0ull;
  }
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < SMALL_EMIT; ++e) surv[((size_t)wv * SMALL_GRID + blockIdx.x) * SMALL_EMIT + e] = best[e];
    bound[(size_t)wv * SMALL_GRID + blockIdx.x] = (unsigned)(best[SMALL_EMIT] >> 32);
  }
}

// 5 <= T <= 16 tokens: the same weight stream on the matrix cores.  One 8-wave workgroup per CU keeps both int8
// planes of the (<= 16) tokens in LDS ([16][d + 16]: the pad spreads the token rows over the banks); a wave owns
// blocks of 16 features n0 .. n0 + 15 (strided over all waves of the grid) and walks k in steps of 64:
//   B = 16 B per lane straight from global (feature l % 16, bytes 16 (l / 16) .. of the step: 16 rows x 64 B per
//       instruction, non-temporal), A = the two planes' fragments from LDS, v_mfma_i32_16x16x64_i8 x 2
// -> C[token 4 (l / 16) + r][feature l % 16] in 4 + 4 accumulator registers.  u = coarse + z sigma as in the dot4
// stream; every lane keeps the SMALL_MF_EMIT + 1 best keys of each of its 4 token slots, the 16 lanes of a token
// group and then the 8 waves merge them (max-reduce rounds), and the workgroup emits its EMIT best + bound.
#ifdef MSAE_MF_PLAIN_LOADS
#define MSAE_MF_LOAD(p) (*(p))
#else
#define MSAE_MF_LOAD(p) __builtin_nontemporal_load(p)
#endif
template <int DSEG>
__global__ __launch_bounds__(512) void gemv_mfma_kernel(const signed char *__restrict__ wqf, const signed char *__restrict__ wqsf,
                                                        const f32x4 *__restrict__ wstat,
                                                        const float *__restrict__ b_enc, int N, int T,
                                                        const signed char *__restrict__ xhi,
                                                        const signed char *__restrict__ xlo,
                                                        const f32x4 *__restrict__ rowc, float zz12, int skip_a,
                                                        int skip_b, unsigned long long *__restrict__ surv,
                                                        unsigned *__restrict__ bound) {
  constexpr int d = DSEG * 1024, PITCH = d + 16, KEEP = SMALL_MF_EMIT + 1, KS = d / 64, UN = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  signed char *xs = reinterpret_cast<signed char *>(smem);                       // [2][16][PITCH]
  unsigned long long *wtop = reinterpret_cast<unsigned long long *>(smem + 2 * 16 * PITCH);   // [16][8][KEEP]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  for (int i = threadIdx.x; i < 2 * 16 * (d / 16); i += 512) {          // planes -> LDS, rows >= T zero
    const int pl = i / (16 * (d / 16)), r = (i / (d / 16)) % 16, c = (i % (d / 16)) * 16;
    i32x4 v = {0, 0, 0, 0};
    if (r < T) v = *reinterpret_cast<const i32x4 *>((pl ? xlo : xhi) + (size_t)r * d + c);
    *reinterpret_cast<i32x4 *>(xs + (size_t)(pl * 16 + r) * PITCH + c) = v;
  }
  float sxz[4], pz[4], rz[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = lg * 4 + r;
    const f32x4 rc = rowc[t < T ? t : T - 1];
    sxz[r] = rc[0]; pz[r] = rc[2]; rz[r] = rc[0] * rc[0] * zz12;
  }
  unsigned long long top[4][KEEP];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < KEEP; ++q) top[r][q] = 0ull;
  __syncthreads();
  const signed char *ah_p = xs + (size_t)l15 * PITCH + lg * 16;          // this lane's A fragment: token l15, k quarter lg
  const signed char *al_p = ah_p + (size_t)16 * PITCH;
  // B fragments from the FRAGMENT-major copies (frag_off: one k-step of a 16-row block = one contiguous kilobyte, this lane's 16 B
  // at byte 16 lane; the row-major copy's 16 rows x 64 B per instruction are half-line requests: 0.13 -> 0.09 ms of stream).  The
  // main copy holds the non-sample rows in main_row order, the sample rows have their own: blocks [0, n_main) | [n_main, N / 16).
  const int n_blocks = N / 16, wave_g = blockIdx.x * 8 + wv, n_waves = gridDim.x * 8;
  const int n_main = MAIN_SKIPS_SAMPLE ? (N - N / SAMPLE_STRIDE) / 16 : n_blocks;
  for (int blk = wave_g; blk < n_blocks; blk += n_waves) {
    const bool samp_blk = blk >= n_main;
    const signed char *bp = (samp_blk ? wqsf + ((size_t)(blk - n_main) * (d / 64) << 10) : wqf + ((size_t)blk * (d / 64) << 10)) + lane * 16;
    i32x4 acc_h = {0, 0, 0, 0}, acc_l = {0, 0, 0, 0};
    i32x4 ba[UN], bb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) ba[u] = MSAE_MF_LOAD(reinterpret_cast<const i32x4 *>(bp + (size_t)u * 1024));
#pragma nounroll
    for (int ks = 0; ks < KS; ks += 2 * UN) {                            // KS % (2 UN) == 0 (d % 1024 == 0)
#pragma unroll
      for (int u = 0; u < UN; ++u) bb[u] = MSAE_MF_LOAD(reinterpret_cast<const i32x4 *>(bp + (size_t)(ks + UN + u) * 1024));
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const i32x4 ah = *reinterpret_cast<const i32x4 *>(ah_p + (ks + u) * 64);
        const i32x4 al = *reinterpret_cast<const i32x4 *>(al_p + (ks + u) * 64);
        acc_h = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, ba[u], acc_h, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, ba[u], acc_l, 0, 0, 0);
      }
      if (ks + 2 * UN < KS) {
#pragma unroll
        for (int u = 0; u < UN; ++u) ba[u] = MSAE_MF_LOAD(reinterpret_cast<const i32x4 *>(bp + (size_t)(ks + 2 * UN + u) * 1024));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const i32x4 ah = *reinterpret_cast<const i32x4 *>(ah_p + (ks + UN + u) * 64);
        const i32x4 al = *reinterpret_cast<const i32x4 *>(al_p + (ks + UN + u) * 64);
        acc_h = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, bb[u], acc_h, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, bb[u], acc_l, 0, 0, 0);
      }
    }
    int n;                                               // feature of this lane's column
    if (samp_blk) n = ((blk - n_main) * 16 + l15) * SAMPLE_STRIDE + SAMPLE_OFF;
    else if (MAIN_SKIPS_SAMPLE) { const int c = blk * 16 + l15, g = c / (SAMPLE_STRIDE - 1), q = c - g * (SAMPLE_STRIDE - 1); n = g * SAMPLE_STRIDE + q + (q >= SAMPLE_OFF ? 1 : 0); }
    else n = blk * 16 + l15;
    const f32x4 st = wstat[n];
    const float bias = b_enc ? b_enc[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = lg * 4 + r;
      const float c = (128.f * (float)acc_h[r] + (float)acc_l[r]) * (sxz[r] * st[0]) + bias;
      const float zs = __builtin_sqrtf(__builtin_fmaf(pz[r], st[1], rz[r] * st[2]));
      unsigned long long key = t < T ? rank_key((n == skip_a || n == skip_b) ? -__builtin_inff() : c + zs, n) : 0ull;
#pragma unroll
      for (int q = 0; q < KEEP; ++q) {                   // sorted insert: the list stays descending
        const unsigned long long cur = top[r][q];
        const bool gt = key > cur;
        top[r][q] = gt ? key : cur;
        key = gt ? cur : key;
      }
    }
  }
  // the 16 lanes of a token group merge their lists: KEEP rounds of "largest head wins and is popped"
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int round = 0; round < KEEP; ++round) {
      unsigned long long m = top[r][0];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 16);
        m = o > m ? o : m;
      }
      if (top[r][0] == m && m != 0ull) {                 // keys are unique: exactly one lane pops
#pragma unroll
        for (int q = 0; q + 1 < KEEP; ++q) top[r][q] = top[r][q + 1];
        top[r][KEEP - 1] = 0ull;
      }
      if (l15 == round) wtop[((size_t)(lg * 4 + r) * 8 + wv) * KEEP + round] = m;
    }
  }
  __syncthreads();
  // wave w finishes tokens 2 w and 2 w + 1: the 8 waves' lists (8 KEEP keys) -> EMIT best + bound
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int t = wv * 2 + half;
    if (t >= T) continue;
    unsigned long long k0 = lane < 8 * KEEP ? wtop[(size_t)t * 8 * KEEP + lane] : 0ull;
    unsigned long long best[KEEP];
#pragma unroll
    for (int e = 0; e < KEEP; ++e) {
      unsigned long long m = k0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
      }
      best[e] = m;
      if (k0 == m) k0 = 0ull;
    }
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < SMALL_MF_EMIT; ++e) surv[(size_t)t * SMALL_SURV + (size_t)blockIdx.x * SMALL_MF_EMIT + e] = best[e];
      bound[(size_t)t * SMALL_GRID + blockIdx.x] = (unsigned)(best[SMALL_MF_EMIT] >> 32);
    }
  }
}

// one 1024-thread workgroup per token, six survivors per thread in registers.  A bisection on the 32-bit
// order key of the upper value finds a threshold with SMALL_R .. SMALL_RMAX survivors at or above it (one
// ballot count + one barrier per step, ~16 steps); those are the candidates (any order), and tau = the largest
// upper value any OTHER feature can have = max(survivors below the threshold, the workgroups' bounds).
// Ties that make the window unreachable leave fewer candidates: still sound, tau says so.
__global__ __launch_bounds__(1024) void select_small_kernel(const unsigned long long *__restrict__ surv,
                                                            const unsigned *__restrict__ bound,
                                                            unsigned long long *__restrict__ cand,
                                                            float *__restrict__ tau, int n_surv, int n_bound) {
  constexpr int PER = SMALL_SURV / 1024;
  static_assert(SMALL_SURV % 1024 == 0 && SMALL_GRID % 1024 == 0, "survivors per thread");
  __shared__ int cnt[33];
  __shared__ unsigned long long c_keys[128];
  __shared__ unsigned s_tau;
  __shared__ int s_n;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  unsigned long long k[PER];
  unsigned v[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    k[e] = e * 1024 + tid < n_surv ? surv[(size_t)t * SMALL_SURV + e * 1024 + tid] : 0ull;
    v[e] = (unsigned)(k[e] >> 32);
  }
  unsigned below = 0u;                                   // largest value that will NOT be a candidate
#pragma unroll
  for (int e = 0; e < SMALL_GRID / 1024; ++e) {
    const unsigned b = e * 1024 + tid < n_bound ? bound[(size_t)t * SMALL_GRID + e * 1024 + tid] : 0u;
    below = b > below ? b : below;
  }
  if (tid < 33) cnt[tid] = 0;
  if (tid < 128) c_keys[tid] = 0ull;
  if (tid == 0) { s_tau = 0u; s_n = 0; }
  __syncthreads();
  unsigned lo = 0u, hi = 0xFFFFFFFFu, theta = 0xFFFFFFFFu;    // f(lo) > SMALL_RMAX, f(hi) < SMALL_R
  for (int step = 0; hi - lo > 1u; ++step) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int c = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v[e] >= mid));
    if (lane == 0) atomicAdd(&cnt[step], c);
    __syncthreads();
    const int tot = cnt[step];
    if (tot > SMALL_RMAX) lo = mid;
    else if (tot < SMALL_R) hi = mid;
    else { theta = mid; break; }
  }
  if (theta == 0xFFFFFFFFu) theta = hi;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    if (v[e] >= theta && k[e] != 0ull) {
      const int slot = atomicAdd(&s_n, 1);
      if (slot < 128) c_keys[slot] = k[e];
    } else {
      below = v[e] > below ? v[e] : below;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(below, off, 64); below = o > below ? o : below; }
  if (lane == 0) atomicMax(&s_tau, below);
  __syncthreads();
  if (tid < 128) cand[(size_t)t * 128 + tid] = c_keys[tid];
  if (tid == 0) {
    const unsigned ninf = f32_order_key(-__builtin_inff());
    tau[t] = f32_from_order_key(s_tau > ninf ? s_tau : ninf);
  }
}

// canonical top-k of a token's exact values (+ the steering hook's set_feature), verification, outputs:
// run by the last rescoring wave of the token
__device__ __forceinline__ void finalize_small(unsigned long long *keys, const unsigned long long *exact, float tau,
                                               int t, int k, int set_feature, float set_value, int viol,
                                               float *vals, IdxOut idx, int32_t *status, int *flagged,
                                               int *n_flagged, int lane) {
  const int has_set = set_feature >= 0 ? 1 : 0;
  for (int i = lane; i < 128; i += 64) {
    unsigned long long kv = 0ull;
    if (i < SMALL_RMAX) kv = exact[(size_t)t * 128 + i];
    else if (has_set) kv = rank_key(set_value, set_feature);
    keys[i] = kv;
  }
  wave_sort_desc_u64<64>(keys, 128, lane);
  const float v_k = f32_from_order_key((unsigned)(keys[k - 1] >> 32));
  const bool ok = (v_k > tau * 1.000001f) && (v_k > 0.f) && (viol == 0);
  for (int j = lane; j < k; j += 64) {
    const unsigned long long key = keys[j];
    const int fi = key ? rank_key_index(key) : 0;
    if (idx.i32) idx.i32[(size_t)t * k + j] = fi;
    if (idx.i64) idx.i64[(size_t)t * k + j] = fi;
    vals[(size_t)t * k + j] = key ? f32_from_order_key((unsigned)(key >> 32)) : 0.f;
  }
  if (lane == 0) {
    if (status) status[t] = ok ? 0 : (2 | (viol ? 64 : 32));
    if (!ok) flagged[atomicAdd(n_flagged, 1)] = t;
  }
}

// grid (SMALL_RMAX, T), one wave each.  Row f of W_enc and the token's activations are loaded straight into
// registers by coalesced 16-B lane loads: lane l holds elements 256 c + 4 l .. + 3 of chunk c.  The exact chain
// is serial by definition; it visits the lanes in order: every lane executes "4 fma, rotate the accumulator
// one lane up" 64 times per chunk, and the lane whose turn it is holds the true partial sum (the others compute
// garbage that is rotated out of the way).  exact[t][r] = rank key of relu(p); a pair further than 6 sigma from
// its coarse value raises viol[t].  The last wave of token t to arrive (device-scope counter) finalises t.
template <int DSEG>
__global__ __launch_bounds__(64) void rescore_small_kernel(const float *__restrict__ a32, const float *__restrict__ W_enc,
                                                           const float *__restrict__ b_enc, int k,
                                                           const unsigned long long *__restrict__ cand,
                                                           const float *__restrict__ tau,
                                                           const f32x4 *__restrict__ wstat,
                                                           const f32x4 *__restrict__ rowc, float zz12, float z2,
                                                           int set_feature, float set_value,
                                                           unsigned long long *__restrict__ exact, int *__restrict__ viol,
                                                           int *__restrict__ done, float *__restrict__ vals, IdxOut idx,
                                                           int32_t *__restrict__ status, int *__restrict__ flagged,
                                                           int *__restrict__ n_flagged) {
  constexpr int d = DSEG * 1024, CH = DSEG * 4, GC = CH < 16 ? CH : 16;   // chunks of 256 elements, <= 16 in registers
  __shared__ unsigned long long keys[128];
  __shared__ int s_last;
  const int r = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
  const unsigned long long ck = cand[(size_t)t * 128 + r];
  if (ck != 0ull) {                                    // wave-uniform
    const int f = rank_key_index(ck);
    const float upper = f32_from_order_key((unsigned)(ck >> 32));
    const float *__restrict__ w = W_enc + (size_t)f * d + lane * 4;
    const float *__restrict__ a = a32 + (size_t)t * d + lane * 4;
    float acc = 0.f;
    for (int g = 0; g < CH; g += GC) {
      f32x4 wv[GC], av[GC];
#pragma unroll
      for (int c = 0; c < GC; ++c) {
        wv[c] = *reinterpret_cast<const f32x4 *>(w + (g + c) * 256);
        av[c] = *reinterpret_cast<const f32x4 *>(a + (g + c) * 256);
      }
#pragma unroll
      for (int c = 0; c < GC; ++c) {
        const f32x4 w4 = wv[c], a4 = av[c];
#pragma unroll 8
        for (int st = 0; st < 64; ++st) {
          acc = __builtin_fmaf(a4[0], w4[0], acc);
          acc = __builtin_fmaf(a4[1], w4[1], acc);
          acc = __builtin_fmaf(a4[2], w4[2], acc);
          acc = __builtin_fmaf(a4[3], w4[3], acc);
          // wave_ror:1 -- lane l takes lane l - 1's value, lane 0 lane 63's
          acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x13C, 0xF, 0xF, false));
        }
      }
    }
    if (lane == 0) {                                   // after whole chunks the true sum is back in lane 0
      const float pre = acc + (b_enc ? b_enc[f] : 0.f);
      exact[(size_t)t * 128 + r] = rank_key(pre > 0.f ? pre : 0.f, f);
      if (upper > -__builtin_inff()) {
        const f32x4 rc = rowc[t], st = wstat[f];
        const float zs2 = __builtin_fmaf(rc[2], st[1], rc[0] * rc[0] * zz12 * st[2]);
        const float diff = pre - (upper - __builtin_sqrtf(zs2));
        if (diff * diff * z2 > GUARD_Z_CHECK * GUARD_Z_CHECK * zs2 * 1.0001f + 1e-30f) atomicOr(viol + t, 1);
      }
    }
  } else if (lane == 0) {
    exact[(size_t)t * 128 + r] = 0ull;
  }
  // release our result, count this wave in; the last one acquires everybody's
  if (lane == 0)
    s_last = __hip_atomic_fetch_add(done + t, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every lane reads the other waves' results below
  const int vi = __hip_atomic_load(viol + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  finalize_small(keys, exact, tau[t], t, k, set_feature, set_value, vi, vals, idx, status, flagged, n_flagged, lane);
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
  const float z = co.z, zz12 = z * z / 12.f;
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
                     flagged, T + 64 + pl.fb_chunks, valid, need, T);
  prof_mark(co.prof, 1, s);
  prof_mark(co.prof, 2, s);
  prof_mark(co.prof, 3, s);
  const int skip_a = set_feature >= 0 ? set_feature : -1, skip_b = zero_feature >= 0 ? zero_feature : -1;
#define MSAE_GEMV(DSEG, TT)                                                                                        \
  hipLaunchKernelGGL((gemv_small_kernel<DSEG, TT>), dim3(SMALL_GRID), dim3(256), 0, s, wq, wstat, b_enc, N, T, xhi, xlo, \
                     rowc, zz12, skip_a, skip_b, surv, bound)
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
                       zz12, skip_a, skip_b, surv, bound);                                                         \
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
                     rowc, zz12, z * z, set_feature, set_value, exact, viol, done, vals, idx, status, flagged, n_flagged)
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
                     pl.i8 ? reinterpret_cast<int *>(ws + pl.off_colmax) : (int *)nullptr, pl.i8 ? (size_t)d * COLMAX_PARTS : (size_t)0);
  if (!pl.i8)
    hipLaunchKernelGGL(prep_x_kernel<DT>, dim3(2048), dim3(256), 0, s, x, b_dec, T, pl.Tp, d, xb, a32);

  GemmOperands op_main{}, op_samp{};
  const float z = co.z, zz12 = z * z / 12.f;
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
    if (shard)   // no re-score on this rank: quantise straight from x - b_dec, a32 is never written
      hipLaunchKernelGGL((quant_x_kernel<DT, true>), dim3(pl.Tp), dim3(256), 0, s, x, b_dec, T, d, odims, is_out, xq, xqo, rowc, zz12, tile_major,
                         valid, need);
    else
      hipLaunchKernelGGL((quant_x_kernel<MSAE_F32, false>), dim3(pl.Tp), dim3(256), 0, s, (const void *)a32, (const float *)nullptr,
                         T, d, odims, is_out, xq, xqo, rowc, zz12, tile_major, valid, need);
    skip_sample = MAIN_SKIPS_SAMPLE && w_packed;   // the tile-major main operand holds the non-sample rows only
    cc_perm = reinterpret_cast<f32x4 *>(ws + pl.off_colc_p);
    hipLaunchKernelGGL(gather_wo_kernel, dim3(N / 32), dim3(256), 0, s, wq, N, d, odims,
                       reinterpret_cast<const f32x4 *>(prepared + pp.off_wstat), wqo, wqos, cc_main, cc_samp, cc_perm,
                       skip_sample ? 1 : 0);
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
    ep.rowc = rowc; ep.colc = colc_s; ep.refs = refs; ep.zz12 = zz12;
#ifdef MSAE_GEMM_TIMELINE
    ep.timeline = nullptr;
#endif
#ifdef MSAE_GEMM_RING64
    const int grc = pl.i8 ? (op_samp.packed == 2 ? gemm64_launch<GemmI8R64, true>(op_samp, T, pl.Tp, pl.S, ep, s)
                                                 : gemm_launch<GemmI8, true>(op_samp, T, pl.Tp, pl.S, ep, s))
                          : gemm_launch<GemmBf16, true>(op_samp, T, pl.Tp, pl.S, ep, s);
#else
    const int grc = skinny == 64    ? gemm_skinny_launch<64, true>(op_samp, T, d, pl.S, ep, s)
                    : skinny == 128 ? gemm_skinny_launch<128, true>(op_samp, T, d, pl.S, ep, s)
                    : skinny == 256 ? gemm_skinny_launch<256, true>(op_samp, T, d, pl.S, ep, s)
                    : pl.i8         ? gemm_launch<GemmI8, true>(op_samp, T, pl.Tp, pl.S, ep, s)
                                    : gemm_launch<GemmBf16, true>(op_samp, T, pl.Tp, pl.S, ep, s);
#endif
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
    ep.rowc = rowc; ep.colc = skip_sample ? cc_perm : colc; ep.refs = refs; ep.zz12 = zz12;
#ifdef MSAE_GEMM_TIMELINE
    if (!g_timeline) (void)hipMalloc(&g_timeline, 64 * 8 * 8);
    (void)hipMemsetAsync(g_timeline, 0, 64 * 8 * 8, s);
    ep.timeline = g_timeline;
#endif
#ifdef MSAE_GEMM_RING64
    const int grc = pl.i8 ? (op_main.packed == 2 ? gemm64_launch<GemmI8R64, false>(op_main, T, pl.Tp, N, ep, s)
                                                 : gemm_launch<GemmI8, false>(op_main, T, pl.Tp, N_main, ep, s))
                          : gemm_launch<GemmBf16, false>(op_main, T, pl.Tp, N, ep, s);
#else
    const int grc = skinny == 64    ? gemm_skinny_launch<64, false>(op_main, T, d, N_main, ep, s)
                    : skinny == 128 ? gemm_skinny_launch<128, false>(op_main, T, d, N_main, ep, s)
                    : skinny == 256 ? gemm_skinny_launch<256, false>(op_main, T, d, N_main, ep, s)
                    : pl.i8         ? gemm_launch<GemmI8, false>(op_main, T, pl.Tp, N_main, ep, s)
                                    : gemm_launch<GemmBf16, false>(op_main, T, pl.Tp, N, ep, s);
#endif
    if (grc) return grc;
  }
  if (pl.segs > 1)
    hipLaunchKernelGGL(compact_candidates_kernel, dim3(T), dim3(64), 0, s, pcnt, pcand, pl.segs, pl.cap, cnt, cand);
  prof_mark(co.prof, 4, s);
  if (shard) {   // feature-sharded group: this shard's best candidates travel, the owner of the token re-scores
    PackArgs pa{};
    pa.cnt = cnt; pa.cand = cand; pa.cap = pl.cap;
    pa.tau_vals = tauv; pa.tau_ld = pl.r; pa.tau_col = pl.r - 1;
    pa.rowc = rowc; pa.colc = colc; pa.zz12 = zz12; pa.i8 = pl.i8 ? 1 : 0;
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
    ra.rowc = rowc; ra.colc = colc; ra.zz12 = zz12; ra.z2 = z * z; ra.i8 = pl.i8 ? 1 : 0;
    ra.set_feature = set_feature; ra.set_value = set_value; ra.zero_feature = zero_feature;
    ra.vals = vals; ra.idx = idx.i32; ra.idx64 = idx.i64; ra.status = status; ra.flagged = flagged; ra.n_flagged = n_flagged;
    ra.fb_cap = T;
    const int nrp = next_pow2(pl.r_max + 1);
    const size_t smem = ((size_t)pl.cap + nrp) * 8 + 64;
    const int lrc = launch_select_rescore<false>(ra, T, k, smem, (const float *)a32, W_enc, s);
    if (lrc) return lrc;
  }
  prof_mark(co.prof, 5, s);
#ifndef MSAE_ABL_NOFALLBACK   // tuning builds only: keep the GEMM ablations' stage timings clean
  rc = run_exact_fallback<DT>(x, W_enc, b_enc, b_dec, T, d, N, k, set_feature, set_value, zero_feature, vals, idx,
                              status, ws, pl, co.detail, s);
  if (rc) return rc;
#endif
  prof_mark(co.prof, 6, s);
  prof_step(co.prof);
  return msae_launch_status();
}

}  // namespace

#ifdef MSAE_GEMM_TIMELINE
extern "C" int msae_debug_timeline(unsigned long long *host_out) {   // tuning builds only (tools/gemm_timeline.py)
  if (!g_timeline) return MSAE_EINVAL;
  return (int)hipMemcpy(host_out, g_timeline, 64 * 8 * 8, hipMemcpyDeviceToHost);
}
#endif

#ifdef MSAE_RESCORE_TL
extern "C" int msae_debug_rescore_timeline(unsigned long long *host_out) {   // tuning builds only (tools/rescore_timeline.py)
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
  opts->reserved = 0;
}

extern "C" int msae_profile_create(int max_steps, void **handle) {
  if (max_steps <= 0 || max_steps > 4096 || !handle) return MSAE_EINVAL;
  ProfState *pf = new ProfState();
  pf->ev = new hipEvent_t[(size_t)max_steps * PROF_MARKS];
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
int prepare_impl(const float *W_enc, int N, int d, void *prepared, int modes, hipStream_t s) {
  if (N <= 0 || d <= 0 || !prepared) return MSAE_EINVAL;
  if (!msae_aligned(prepared, 256)) return MSAE_EALIGN;
  Prepared p = make_prepared(N, d);
  // what this call rebuilds is valid, everything else is stale from now on (the weights have changed)
  p.valid = ((modes & 1) ? PREP_BF16 : 0u) | (((modes & 2) && i8_shape_ok(N, d)) ? (PREP_I8 | ((modes & 4) ? 0u : PREP_FRAG)) : 0u);
  MSAE_HIP_TRY(hipMemcpyAsync(prepared, &p, sizeof(p), hipMemcpyHostToDevice, s));
  if (p.S) {
    if (!msae_aligned(W_enc, 16)) return MSAE_EALIGN;
    unsigned char *base = static_cast<unsigned char *>(prepared);
    if (modes & 1)
      hipLaunchKernelGGL(prepare_weights_kernel, dim3(4096), dim3(256), 0, s, W_enc, N, d,
                         reinterpret_cast<unsigned short *>(base + p.off_wb),
                         reinterpret_cast<unsigned short *>(base + p.off_ws));
    f32x4 *wstat = reinterpret_cast<f32x4 *>(base + p.off_wstat), *wstat_s = reinterpret_cast<f32x4 *>(base + p.off_wstat_s);
    f32x4 *colbf = reinterpret_cast<f32x4 *>(base + p.off_colbf), *colbf_s = reinterpret_cast<f32x4 *>(base + p.off_colbf_s);
    if ((modes & 2) && i8_shape_ok(N, d))   // row statistics (both passes' error bands) + int8 operands
      hipLaunchKernelGGL(row_stats_quant_kernel<true>, dim3(N), dim3(256), 0, s, W_enc, N, d, wstat, wstat_s, colbf,
                         colbf_s, reinterpret_cast<signed char *>(base + p.off_wq),
                         reinterpret_cast<signed char *>(base + p.off_wqs), reinterpret_cast<signed char *>(base + p.off_wqp),
                         reinterpret_cast<signed char *>(base + p.off_wqsp),
                         (modes & 4) ? (signed char *)nullptr : reinterpret_cast<signed char *>(base + p.off_wqf),
                         (modes & 4) ? (signed char *)nullptr : reinterpret_cast<signed char *>(base + p.off_wqsf), gemm_layout() == 2 ? 2 : 1);
    else
      hipLaunchKernelGGL(row_stats_quant_kernel<false>, dim3(N), dim3(256), 0, s, W_enc, N, d, wstat, wstat_s, colbf,
                         colbf_s, (signed char *)nullptr, (signed char *)nullptr, (signed char *)nullptr, (signed char *)nullptr,
                         (signed char *)nullptr, (signed char *)nullptr, 1);
  }
  return msae_launch_status();
}
}  // namespace

extern "C" int msae_encoder_prepare(const float *W_enc, int N, int d, void *prepared, void *stream) {
  return prepare_impl(W_enc, N, d, prepared, 3, (hipStream_t)stream);
}

// After a weight update (training): rebuild only the operands the coarse mode in force reads.
extern "C" int msae_encoder_refresh(const float *W_enc, int N, int d, void *prepared, const msae_options *opts,
                                    void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co)) return MSAE_EINVAL;
  const bool i8 = co.mode == 1 && i8_shape_ok(N, d);
  return prepare_impl(W_enc, N, d, prepared, i8 ? 2 : 1, (hipStream_t)stream);
}

// ... for an encode of T_next tokens that follows: a batch of more than 256 tokens does not read the fragment-major copies (0.5 GB
// of scattered 16-byte stores per refresh at C2).  The buffer must be refreshed again before an encode of fewer tokens.
extern "C" int msae_encoder_refresh_for(const float *W_enc, int N, int d, void *prepared, int T_next, const msae_options *opts,
                                        void *stream) {
  CallOpts co;
  if (!resolve_opts(opts, co) || T_next <= 0) return MSAE_EINVAL;
  const bool i8 = co.mode == 1 && i8_shape_ok(N, d);
  return prepare_impl(W_enc, N, d, prepared, (i8 ? 2 : 1) | (T_next > 256 ? 4 : 0), (hipStream_t)stream);
}

extern "C" size_t msae_encode_topk_ws_bytes(int T, int d, int N, int k, const msae_options *opts) {
  if (T <= 0 || d <= 0 || N <= 0 || k <= 0) return 0;
  CallOpts co;
  if (!resolve_opts(opts, co)) return 0;
  const FusedPlan pl = make_plan(T, d, N, k, co.mode);
  return pl.bytes;
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
  FusedPlan pl = make_plan(T, d, N, k, co.mode);
  if (!prepared && pl.fast) return MSAE_EINVAL;  // the fast path needs msae_encoder_prepare()
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
  if (!pl.fast || !prepared || C > pl.cap) return MSAE_ENOTIMPL;   // shapes without the candidate pass: use msae_encode_topk per shard
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
