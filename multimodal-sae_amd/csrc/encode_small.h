// encode_small.h -- the small-T path of the fused encoder (T <= 16: steering decode steps, features/steering.py:86,105-124):
// weight-stream kernels (dot4 and MFMA), threshold select, one-wave-per-pair exact re-score with in-kernel finalisation.
// Host dispatch: encode_fused.hip (run_small).
#pragma once
#include "encode_defs.h"

namespace {

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

// ---- small-T path kernels ---------------------------------------------------------------------------------
// one 256-thread workgroup per token: a32, two-plane quantisation q = rint(a / s), q = 128 hi + lo with
// hi in [-127, 127], lo in [-64, 63], s = max|a| / 16319; rowc[t] = (s, 1, z^2 |a|^2 / 12, 0)
template <int DT>
__global__ __launch_bounds__(256) void prep_small_kernel(const void *__restrict__ x, const float *__restrict__ b_dec,
                                                         int d, float *__restrict__ a32, signed char *__restrict__ xhi,
                                                         signed char *__restrict__ xlo, f32x4 *__restrict__ rowc,
                                                         float zz12, int *__restrict__ zero_a, int n_a,
                                                         int *__restrict__ zero_b, int n_b,
                                                         const unsigned *__restrict__ valid, unsigned need, int T,
                                                         unsigned long long seed) {
  __shared__ float red[2][4];
  const int t = blockIdx.x;
  const bool dith = seed != 0ull;                      // msae_options::dither (quant_x_kernel)
  const unsigned dkey = dith ? dither_key(seed, (unsigned)t) : 0u;
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
  // (dither: the weights' residuals multiply the DEQUANTISED activation, |A + delta| <= |A| + 1 step per dim -- the cross term of
  // the two roundings inside the Hoeffding proxy: P = z^2 (|a| + s sqrt(d))^2 / 12, x 3 in Q_n)
  const float an = __builtin_sqrtf(ss) + (dith ? scale * __builtin_sqrtf((float)d) : 0.f);
  if (threadIdx.x == 0) rowc[t] = f32x4{scale, 1.f, zz12 * an * an, 0.f};
  const float inv = 1.f / scale;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(a32 + (size_t)t * d + c);
    unsigned wh = 0, wl = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q = dith ? (int)floorf(v[e] * inv + dither01(dkey, (unsigned)(c + e))) : (int)rintf(v[e] * inv);
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
                                                           const f32x4 *__restrict__ rowc, float zz12, float z2, float zc2,
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
        if (diff * diff * z2 > zc2 * zs2 * 1.0001f + 1e-30f) atomicOr(viol + t, 1);
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

}  // namespace
