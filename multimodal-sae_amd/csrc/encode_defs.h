// encode_defs.h -- constants, the prepared-encoder layout, per-call options and the error-band arithmetic shared by the kernels
// of the fused encoder (encode_prep.h, encode_rescore.h, encode_small.h) and its host dispatch (encode_fused.hip).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <chrono>
#include <random>

#include "common.h"
#include "tuning.h"

namespace {

constexpr int SAMPLE_STRIDE = 32, SAMPLE_OFF = 13;
// Tokens one pass of the in-call exact fallback absorbs: its dense scratch rows are budgeted at 4 GiB
// (8192 rows at N = 131072; 1 GiB until round 5: the bench batch then enqueued four passes = eight empty
// launches + a count kernel per step, ~30 us of a 5.9 ms step, for scratch nobody touches unless a token is
// flagged -- 288 GB of HBM make the other trade), never fewer than 128 and never more than the call has tokens.  The
// exact kernels take the flagged count from device memory and loop over it; ceil(T / capacity) passes
// are enqueued (the ones without work exit at once), so EVERY flagged token is recomputed inside the
// call whatever their number -- no host round trip, no "unresolved" leftovers.
constexpr size_t FB_BUDGET_BYTES = (size_t)4 << 30;
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
  // Round 6, subtractive dither of the large-batch int8 pass (sd_* below): the seed the int8 operands of this buffer were rounded
  // with (0: round to nearest) -- the activations of a batch are rounded against vectors derived from the SAME seed, so that the
  // per-feature correction D_n can be computed where the row is quantised -- and off_ds: Ds[n] = sw_n * D_n, f32 [N].
  unsigned long long dseed;
  size_t off_ds;
  // the two dither vectors as a table, h_x(c) << 16 | h_w(c) int32 [d] (sd_h16 of the two keys), and behind it F = sum_c g_w(c) g_x(c)
  // (int64, 2^-34 units): written by sd_table_kernel in front of every kernel that quantises rows with the seed
  size_t off_sdtab;
};
constexpr unsigned PREP_MAGIC = 0x4D534145u;  // "MSAE"
constexpr unsigned PREP_BF16 = 1u;   // W_bf16 + bf16 sample rows
constexpr unsigned PREP_I8 = 2u;     // Wq row-major, tile-major (+ sample copies)
constexpr unsigned PREP_FRAG = 4u;   // Wq fragment-major (+ sample copy): the weight-stream kernels of <= 128 tokens
constexpr unsigned PREP_F8 = 8u;     // fp8 (e4m3) operands, tile-major: ALL rows in the row-major int8 region (off_wq), the sample rows in off_wqsp
                                     // -- they overwrite int8 operands, so PREP_I8 / PREP_FRAG and PREP_F8 exclude each other

__host__ __device__ inline bool fast_shape_ok(int N, int d) {
  return N % (SAMPLE_STRIDE * 256) == 0 && d % 64 == 0;  // sample width N/32 must tile by BN = 256
}
__host__ __device__ inline bool i8_shape_ok(int N, int d) { return fast_shape_ok(N, d) && d % 128 == 0; }

// The tile-major operand of the main candidate pass leaves the sample rows out (the sample pass has scored them: their
// candidates are taken from its output, sample_push_kernel): -1/32 of the pass's matrix work and operand traffic.  Row n of W
// (n not a sample row) is row main_row(n) of that operand; column c of the pass is feature gemm_feature(c).  31/32 N tiles by
// 256 whenever the sample width does (fast_shape_ok).  (tuning.h: MSAE_FULL_MAIN_PASS keeps all rows in.)
constexpr bool MAIN_SKIPS_SAMPLE = msae_tuning::MAIN_SKIPS_SAMPLE;
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
  p.off_ds = take(q ? (size_t)N * 4 : 0);
  p.off_sdtab = take(q ? (size_t)d * 4 + 24 : 0);
  p.bytes = o;
  return p;
}

// ---- per-call options (msae_options, include/msae.h), resolved once per entry-point call.  The library holds no
// mutable state: the environment only supplies DEFAULTS (read at the call, never cached), everything else travels
// with the call.
struct ProfState;
struct CallOpts {
  int mode;          // coarse-pass operand type: 0 = bf16, 1 = int8, 2 = fp8 (e4m3)
  float z;           // width of the error band: u = coarse + z*sigma
  int detail;        // status = 1 | reason << 8 for tokens recomputed in the call
  ProfState *prof;   // stage timing handle or null
  int exact;         // every token by the exact path (msae_options::exact)
  int32_t *rows_out; // per-token re-score statistics (msae_options::rows_rescored) or null
  // Stochastic rounding of the int8 operands (msae_options::dither): seed != 0 <=> on.  The seed of THIS call: the caller's, or
  // drawn here (draw_seed) -- activations are rounded with it in an encode, the weights in a prepare / refresh.
  unsigned long long seed;
  int cert;              // msae_options::certified: two-plane operands, deterministic band (encode_cert.h)
  const void *cert_ops;  // ... and msae_encoder_prepare_certified's buffer
};
// 64-bit finaliser of splitmix64 (also the device-side hash of the dither, encode_prep.h)
__host__ __device__ inline unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// A fresh non-zero seed: process-random base (std::random_device at first use) + an atomic counter, mixed.  The only
// mutable process state of the library -- it is a random number generator.
inline unsigned long long draw_seed() {
  static const unsigned long long base = [] {
    unsigned long long b = 0x6D736165ull;
    try {
      std::random_device rd;
      b ^= ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
    } catch (...) {   // no entropy source: clock and address-space layout (an exception must not cross the C ABI)
      b ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ (unsigned long long)(size_t)&b;
    }
    return b;
  }();
  static std::atomic<unsigned long long> counter{0};
  const unsigned long long v = mix64(base + 0x9E3779B97F4A7C15ull * (counter.fetch_add(1, std::memory_order_relaxed) + 1ull));
  return v ? v : 1ull;
}
inline bool resolve_opts(const msae_options *o, CallOpts &c) {
  c.mode = -1; c.z = 0.f; c.detail = 0; c.prof = nullptr; c.exact = 0; c.rows_out = nullptr; c.seed = 0ull;
  c.cert = 0; c.cert_ops = nullptr;
  int dither = 0;
  if (o) {
    // `size` is the caller's sizeof: a caller compiled against ABI 2's header (no `exact`) is served with exact = 0
    if (o->size < offsetof(msae_options, exact)) return false;
    c.mode = o->coarse_mode; c.z = o->guard_z; c.detail = o->status_detail ? 1 : 0;
    c.prof = static_cast<ProfState *>(o->profile);
    if (o->size >= offsetof(msae_options, exact) + sizeof(int32_t)) c.exact = o->exact ? 1 : 0;
    if (o->size >= offsetof(msae_options, dither) + sizeof(int32_t)) dither = o->dither;
    if (o->size >= offsetof(msae_options, rows_rescored) + sizeof(void *)) c.rows_out = o->rows_rescored;
    if (o->size >= offsetof(msae_options, dither_seed) + sizeof(uint64_t)) c.seed = o->dither_seed;
    if (o->size >= offsetof(msae_options, certified_operands) + sizeof(void *)) {
      c.cert = o->certified ? 1 : 0;
      c.cert_ops = o->certified_operands;
    }
  }
  if (dither < 0 || dither > 2) return false;
  if (dither == 0) {
    const char *e = getenv("MSAE_DITHER");
    dither = (e && e[0] == '0') ? 2 : 1;
  }
  if (dither == 2) c.seed = 0ull;
  else if (c.seed == 0ull) c.seed = draw_seed();
  if (c.mode < 0) {
    const char *e = getenv("MSAE_COARSE");
    c.mode = (e && e[0] == 'b') ? 0 : ((e && e[0] == 'f') ? 2 : 1);
  }
  if (c.mode != 0 && c.mode != 1 && c.mode != 2) return false;
  if (c.z == 0.f) {
    const char *e = getenv("MSAE_GUARD_Z");
    const float v = e ? (float)atof(e) : 0.f;
    c.z = (v >= 0.25f && v <= 64.f) ? v : 7.f;
  }
  return c.z >= 0.25f && c.z <= 64.f;
}

unsigned long long *g_timeline = nullptr;   // tuning builds only (msae_tuning::GEMM_TIMELINE)
// Model check: a re-scored pair further than this many sigma (of the band's sigma) from its coarse value flags the token.
// Round to nearest: 6 sigma of the noise model (2e-9 per pair under it) -- the data-side assumption is being tested.  Dither:
// the band's sigma is Hoeffding's proxy, sqrt(3/2) x the actual rms on fine data, and nothing about the data is assumed any more
// (the check is only the net under operands edited behind the API): 5 proxy sigma = 6.1 actual, the same false-alarm rate.
__host__ __device__ inline float guard_z_check2(bool dither) { return dither ? 25.f : 36.f; }
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
// x-side variance of one rounding in the band, in steps^2: 1/12 (round to nearest, the statistical model) or, under the dither,
// Hoeffding's variance proxy of a zero-mean term bounded by one step, 1/4 -- times 1.001, which covers the 2^-25 granularity of
// the hash's uniform (a bias of <= 2^-25 step per dim) and the float rounding of v / step at the clamp.  The W side carries the
// same factor inside Q_n (row_stats_quant_row).
__host__ __device__ inline float x_round_var(bool dither) { return dither ? 0.25f * 1.001f : 1.f / 12.f; }
constexpr float GUARD_ZETA = MSAE_GUARD_ZETA;   // first round reaches zeta sigma below the k-th coarse value
constexpr float BF16_REL_VAR2 = 5.5e-6f;  // variance of the sum of two relative bf16 roundings (2 x 2^-16/3 x E[1/m^2])
// fp8 (e4m3: 4 significant bits, half an ulp = 2^-4 of the binade): 2 x 2^-8/3 x E[1/m^2] = 256 x the bf16 figure.  Elements below
// the format's normal range round on an ABSOLUTE grid instead: after the per-token / per-feature scaling (largest element -> 224,
// inside OCP e4m3's 448 and FNUZ's 240) the subnormal step is at most 2^-9 of a scaled unit; their variance enters the band as
// sx^2 |W_n|^2 v_abs + sw_n^2 |a_t|^2 v_abs, v_abs = 2^-18 / 3 (a factor 4 of slack on (2^-10)^2 / 3 for either format).
constexpr float FP8_REL_VAR2 = 256.f * 5.5e-6f;
constexpr float FP8_ABS_VAR = 3.8146973e-6f / 3.f;
constexpr float FP8_MAX = 224.f;

// index output of one call: 32-bit (msae_encode_topk), 64-bit (msae_encode_topk_i64), never both null
struct IdxOut { int32_t *i32; int64_t *i64; };

// z^2 sigma^2 of one (token, feature) pair; rc = (sx, m, P, -), cc = (sw, Q, Si, So).  Same expression as the
// GEMM epilogue (gemm_mfma.h).  zz12 = z^2 x the x-side variance of one rounding (x_round_var: 1/12, or 1/4 under the dither);
// the W side's is inside P_t Q_n (P = z^2 |a|^2 / 12, Q = sw^2 or 3 sw^2).
__device__ __forceinline__ float band_sq(const f32x4 rc, const f32x4 cc, float zz12, bool i8) {
  if (!i8) return rc[2] * cc[1];
  const float rz = rc[0] * rc[0] * zz12;
  return __builtin_fmaf(rc[2], cc[1], __builtin_fmaf(rz * rc[1] * rc[1], cc[3], rz * cc[2]));
}

// Dither of the activations (quant_x_kernel, prep_small_kernel): one 32-bit hash per element, keyed per token.
//   key  = low word of mix64(seed + golden * (t + 1))      (once per thread and token)
//   r(c) = (lowbias32(key ^ c * 0x9E3779B9) >> 8 + 0.5) * 2^-24   in (0, 1)
// Independence is needed across the dims of ONE token only (the Hoeffding sum of a (token, feature) pair runs over c).
__device__ __forceinline__ unsigned dither_key(unsigned long long seed, unsigned t) {
  return (unsigned)mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1u));
}
__device__ __forceinline__ float dither01(unsigned key, unsigned c) {
  unsigned h = key ^ (c * 0x9E3779B9u);
  h ^= h >> 16; h *= 0x7FEB352Du;
  h ^= h >> 15; h *= 0x846CA68Bu;
  h ^= h >> 16;
  return ((float)(h >> 8) + 0.5f) * (1.f / 16777216.f);
}
// ---- subtractive dither (round 6) ------------------------------------------------------------------
// Non-subtractive stochastic rounding q = floor(A + r) leaves a residual of variance f(1 - f) <= 1/4 step^2 that depends on the
// input, so the every-input band had to use Hoeffding's worst case 1/4 where round to nearest has 1/12 (a sqrt(3) wider band:
// 58 rows re-scored per token instead of 45).  With the dither SUBTRACTED again -- the operand element is taken as q - (r - 1/2)
// -- the residual delta = q - (r - 1/2) - A is EXACTLY uniform on (-1/2, 1/2] whatever A is (Schuchman), independent across the
// dims whose r are independent, and a uniform variable is sub-Gaussian with variance proxy equal to its variance:
// E exp(l delta) = sinh(l/2) / (l/2) <= exp(l^2 / 24).  The Chernoff bound of the coarse value's error then holds with 1/12 per
// rounding, for every input, over the library's randomness alone.  Subtracting costs a correction per output,
//   sum_c (q_c - dx_c)(w^_c - dw_c) = sum q w^  -  D_n  -  E_t + F,   D_n = sum_c dx_c w^_nc,  E_t = sum_c q_tc dw_c,  F = sum dx dw,
// which is affordable only when the dither is SHARED: dx_c the same for all tokens, dw_c the same for all features (the proof
// needs independence across the dims c of ONE (token, feature) pair only).  D_n is then a per-feature constant computed where row
// n is quantised (row_stats_quant_row), E_t - F a per-token integer out of quant_x_kernel, and the candidate GEMM applies both
// for free: E in the multiply-add that scales the outlier tile (acc = acc * m - E), D inside the epilogue's fma nesting
// ((acc * sw - sw D) * sx + b).  Both vectors derive from the seed of the prepare / refresh (Prepared::dseed).
//   r(c) = (2 h(c) + 1) 2^-17 in (0, 1),  h = 16 hash bits;  d(c) = r(c) - 1/2 = g(c) 2^-17,  g = 2 h + 1 - 2^16  (exact integers:
//   D and E are integer sums -- int32 per thread, int64 across threads -- with no rounding until the final conversion)
__host__ __device__ inline unsigned sd_key(unsigned long long seed, unsigned side) {   // side 0: activations (dx), 1: weights (dw)
  return (unsigned)mix64(seed ^ (0x5D17E5ull + 0x9E3779B97F4A7C15ull * (unsigned long long)(side + 1u)));
}
__host__ __device__ __forceinline__ int sd_h16(unsigned key, unsigned c) {
  unsigned h = key ^ (c * 0x9E3779B9u);
  h ^= h >> 16; h *= 0x7FEB352Du;
  h ^= h >> 15; h *= 0x846CA68Bu;
  h ^= h >> 16;
  return (int)(h >> 16);
}
__host__ __device__ __forceinline__ float sd_r(int h16) { return (float)(2 * h16 + 1) * (1.f / 131072.f); }   // exact
__host__ __device__ __forceinline__ int sd_g(int h16) { return 2 * h16 + 1 - (1 << 16); }
constexpr double SD_UNIT = 1.0 / 131072.0;       // one unit of g
// one table word per dim: h_x(c) << 16 | h_w(c)
__host__ __device__ __forceinline__ int sd_hx(int word) { return (int)((unsigned)word >> 16); }
__host__ __device__ __forceinline__ int sd_hw(int word) { return word & 0xFFFF; }
// Slack on the variances of the subtractive band (times z^2 / 12):  1.001 for the float roundings of A = v / step;
// (1 + 0.01 / z)^2 for the integer rounding of E_t - F (<= 1/2 accumulator unit against z sigma >= z sqrt(2 127^2 / 12) units:
// the largest element of either operand quantises to +-127);  (1 + 2.6e-5 sqrt(d) / z)^2 for the float evaluation of the coarse
// value itself (int32 -> f32 conversion and two roundings, <= 2e-7 |a . w|, against sigma >= |a| |w| / (127 sqrt(12 d)));
// (1 + 2.7e-5 sqrt(d) / z)^2 for the 2^-16 grid of r (a uniform residual on a grid carries a mean of <= 2^-17 step per dim:
// coherently at most sqrt(d) |w| 2^-17 = 2.64e-5 sqrt(d) sigma).
inline float sd_slack(float z, int d) {
  const float e = 1.f + (0.01f + 5.3e-5f * __builtin_sqrtf((float)d)) / z;
  return 1.001f * e * e;
}
// (h_x, h_w) of every dim and F = sum_c g_w(c) g_x(c) (int64, 2^-34 units) into the prepared buffer (one workgroup; d is small)
__global__ __launch_bounds__(1024) void sd_table_kernel(unsigned long long seed, int d, int *__restrict__ tab) {
  __shared__ long long red[16];
  const unsigned kx = sd_key(seed, 0u), kw = sd_key(seed, 1u);
  long long f = 0;
  for (int c = threadIdx.x; c < d; c += 1024) {
    const int hx = sd_h16(kx, (unsigned)c), hw = sd_h16(kw, (unsigned)c);
    tab[c] = (int)(((unsigned)hx << 16) | (unsigned)hw);
    f += (long long)sd_g(hx) * sd_g(hw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    *reinterpret_cast<long long *>(tab + ((d + 1) & ~1)) = t;
  }
}
// Largest outlier multiplier whose remainder still fits the token's own column: A = m hi + lo, |lo| <= m / 2 + 1 <= 127.
constexpr int SD_M_EXACT = 252;

// ---- per-row statistics + int8 operands ---------------------------------------------------------------
// Tile-major int8 operand of the candidate GEMM (GemmOperands::packed): byte offset of the 16-B chunk at column c
// (c % 16 == 0) of row r, with the LDS image's chunk permutation applied (gemm_swz).  d % 128 == 0.
__host__ __device__ __forceinline__ size_t packed_off(size_t r, int c, int d, int layout = 1) {
  (void)layout;
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
// row-major (environment MSAE_GEMM_ROWMAJOR=1, for A/B runs).  Read at every call: an immutable property of the process
// environment (prepare and encode must agree).
inline int gemm_layout() { return getenv("MSAE_GEMM_ROWMAJOR") ? 0 : 1; }

}  // namespace
