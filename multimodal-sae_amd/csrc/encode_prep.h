// encode_prep.h -- kernels in front of the candidate GEMM of the fused encoder: weight preparation (bf16 copy, per-row
// statistics, int8 operands in three layouts), per-call activation preparation (a32 = x - b_dec, column maxima, outlier
// dims, per-token int8 quantisation + band constants, outlier tile of Wq), the sample features' candidates, list compaction.
// Host dispatch: encode_fused.hip.
#pragma once
#include "encode_defs.h"

namespace {

constexpr int MAX_OUT = 128;   // outlier dims of a batch (pick_outliers_kernel): they fit one int8 k-tile

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

// W side (once per weight load), one 256-thread workgroup per row:
//   sw[n] = max|W[n][:]| / 127,  |W_n|^2,  |W_n|_4^2 = sqrt(sum w^4)   -> wstat[n] = (sw, Q_i8, |W_n|^2, Q_bf)
//   Wq[n][c] = rint(W[n][c] / sw[n])   (QUANT; d % 128 == 0)
// A row whose rms lies below one step (max > 127 rms: its bulk quantises to 0, +-1) would leave a
// STRUCTURED residual (the bulk itself), so such rows are rounded stochastically with a hash dither:
// floor(s + r(n, c)), r uniform in [0, 1) -- unbiased for any activation direction, residual variance
// <= 1/4 step^2 instead of 1/12: Q_i8 = 3 sw^2 for them.
__device__ __forceinline__ float hash01(unsigned long long z) {
  z = mix64(z);
  return (float)(unsigned)(z >> 40) * (1.f / 16777216.f);
}
// The work of one 256-thread workgroup on row n (also the tail of the fused optimiser pass of csrc/train.hip, which calls it
// on the row it has just updated: the operands of the NEXT step's encode without another sweep over W_enc).
struct RowQuantOut {
  f32x4 *wstat, *wstat_s, *colbf, *colbf_s;
  signed char *wq, *wqs, *wqp, *wqsp, *wqf, *wqsf;
  int layout;
  // msae_options::dither of the prepare / refresh: != 0 rounds EVERY row stochastically with this seed (Q_i8 = 3 sw^2 x 1.001:
  // the residual of every weight is the library's randomness); 0: round to nearest, sub-step rows with the fixed hash (ABI 3).
  // Round 6: the dither r(c) is shared by all rows (sd_key(seed, 1); encode_defs.h) -- a consumer that does not subtract it again
  // sees the stochastic rounding it always saw (Hoeffding proxy 1/4, Q_i8), one that does (the large-batch pass) a uniform
  // residual of variance 1/12 -- and ds[n] = sw_n * D_n, D_n = sum_c dx(c) Wq[n][c], is the correction of the ACTIVATIONS' shared
  // dither for this row (sd_key(seed, 0)); 0 without dither.
  unsigned long long seed;
  float *ds;
  const int *sdtab;    // h_x(c) << 16 | h_w(c) of the seed (sd_table_kernel); read when seed != 0
};
template <bool QUANT>
__device__ __forceinline__ void row_stats_quant_row(const float *__restrict__ W, int n, int d, const RowQuantOut &o,
                                                    float (&red)[3][4]) {
  f32x4 *__restrict__ wstat = o.wstat, *__restrict__ wstat_s = o.wstat_s, *__restrict__ colbf = o.colbf, *__restrict__ colbf_s = o.colbf_s;
  signed char *__restrict__ wq = o.wq, *__restrict__ wqs = o.wqs, *__restrict__ wqp = o.wqp, *__restrict__ wqsp = o.wqsp;
  signed char *__restrict__ wqf = o.wqf, *__restrict__ wqsf = o.wqsf;
  const int layout = o.layout;
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
  const bool dither = o.seed != 0ull || s2 < scale * scale * (float)d;       // every row / rms below one step
  const bool shared = o.seed != 0ull;                                        // (sub-step rows of a round-to-nearest prepare: the fixed per-element hash)
  int dacc = 0;                                                              // 16 dims' share of D_n in units of 2^-17 (< 2^27: flushed per iteration)
  long long dacc64 = 0;
  const bool samp = (n % SAMPLE_STRIDE) == SAMPLE_OFF;
  if (threadIdx.x == 0) {
    const float q_bf = __builtin_sqrtf(s4);
    const f32x4 st = {scale, scale * scale * (dither ? 12.f * x_round_var(true) : 1.f), s2, q_bf};
    // bf16 pass: only Q_bf is read.  fp8 pass (quant_w_fp8_kernel / GemmCfg::F8): (sw8, Q_bf, |W_n|^2, sw8^2) -- the value scale and
    // the two absolute-grid terms of its band (encode_defs.h: FP8_ABS_VAR)
    const float sw8 = m > 0.f ? m / FP8_MAX : 1.f;
    const f32x4 cb = {sw8, q_bf, s2, sw8 * sw8};
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
        i32x4 tb = {0, 0, 0, 0};                              // h_x << 16 | h_w of the four dims
        if (shared) tb = *reinterpret_cast<const i32x4 *>(o.sdtab + c + 4 * q);
        unsigned w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = v[e] * inv;
          const unsigned cc = (unsigned)(c + 4 * q + e);
          int iv = shared   ? (int)floorf(sv + sd_r(sd_hw(tb[e])))
                   : dither ? (int)floorf(sv + hash01((unsigned long long)n * (unsigned)d + cc))
                            : (int)rintf(sv);
          iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
          if (shared) dacc += sd_g(sd_hx(tb[e])) * iv;
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
      dacc64 += dacc; dacc = 0;                  // (any d: the 32-bit partial never holds more than 16 products)
    }
    if (o.ds) {                                  // D_n: int64 sum over the workgroup (red[] was consumed above)
      long long dsum = dacc64;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
      __syncthreads();
      unsigned *r32 = reinterpret_cast<unsigned *>(&red[0][0]);
      if ((threadIdx.x & 63) == 0) { r32[2 * (threadIdx.x >> 6)] = (unsigned)dsum; r32[2 * (threadIdx.x >> 6) + 1] = (unsigned)((unsigned long long)dsum >> 32); }
      __syncthreads();
      if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < 4; ++w) t += (long long)(((unsigned long long)r32[2 * w + 1] << 32) | r32[2 * w]);
        o.ds[n] = scale * (float)((double)t * SD_UNIT);
      }
    }
  }
}
template <bool QUANT>
__global__ __launch_bounds__(256) void row_stats_quant_kernel(const float *__restrict__ W, int N, int d, RowQuantOut o) {
  __shared__ float red[3][4];
  row_stats_quant_row<QUANT>(W, blockIdx.x, d, o, red);
}

// ---- fp8 (e4m3) operands (BASELINE configs[4]) ----------------------------------------------------------------------------
__device__ __forceinline__ int pack4_fp8(const f32x4 v) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  return w;
}
// W side, once per weight load: row n scaled by 1 / sw8 (colbf[n][0], written by row_stats_quant_kernel in front of this) -> e4m3,
// tile-major: every row into `w8` (full width, row n), the sample rows also into `w8s`.  One 256-thread workgroup per row.
__global__ __launch_bounds__(256) void quant_w_fp8_kernel(const float *__restrict__ W, int N, int d,
                                                         const f32x4 *__restrict__ colbf, signed char *__restrict__ w8,
                                                         signed char *__restrict__ w8s) {
  const int n = blockIdx.x;
  const float inv = 1.f / colbf[n][0];
  const bool samp = (n % SAMPLE_STRIDE) == SAMPLE_OFF;
  const float *row = W + (size_t)n * d;
  for (int c = threadIdx.x * 16; c < d; c += 4096) {
    i32x4 packed;
#pragma unroll
    for (int q = 0; q < 4; ++q) packed[q] = pack4_fp8(*reinterpret_cast<const f32x4 *>(row + c + 4 * q) * inv);
    *reinterpret_cast<i32x4 *>(w8 + packed_off((size_t)n, c, d)) = packed;
    if (samp) *reinterpret_cast<i32x4 *>(w8s + packed_off((size_t)(n / SAMPLE_STRIDE), c, d)) = packed;
  }
}
// x side, every call: one 256-thread workgroup per token row of the padded tile.  a32 holds x - b_dec; is_out / odims are the
// batch's massive-activation dims (pick_outliers_kernel, as on the int8 path).  The band of a pair (encode_defs.h) is
//   z^2 sigma^2 = z^2 v_rel [ |a_in|_4^2 |W_n|_4^2 + max_out(a^2) sum_out W_n[c]^2 ] + z^2 v_abs [ sx^2 |W_n|^2 + sw_n^2 |a|^2 ]
// -- Cauchy-Schwarz on sum a_c^2 w_c^2 over the ordinary dims only: with the handful of x20 .. x1000 dims inside, |a|_4^2 is ~15x
// the ordinary dims' and the band swallows hundreds of features (first build of this pass: 95 % of the tokens past r_max) --
// in the three-term form  P_t Q_n + R_t (Si_n + M_t^2 So_n),  R_t = sx^2 z^2 v_abs:
//   rowc[t] = (sx = max|a| / 224,  M = sqrt(v_rel / v_abs) max_out|a| / sx,  P = z^2 v_rel |a_in|_4^2,  flag)
//   colc[n] = (sw8, Q = |W_n|_4^2 + (v_abs / v_rel) sqrt(d) sw8^2,  Si = |W_n|^2,  So = sum_out W_n[c]^2 + (v_abs / v_rel) n_out sw8^2)
// (the W-side absolute term sw^2 |a|^2 <= sw^2 (sqrt(d) |a_in|_4^2 + n_out max_out a^2) rides in Q and So).
__global__ __launch_bounds__(256) void quant_x_fp8_kernel(const float *__restrict__ a32, int T, int d,
                                                         const unsigned char *__restrict__ is_out, signed char *__restrict__ x8,
                                                         f32x4 *__restrict__ rowc, float z2, const unsigned *__restrict__ valid) {
  __shared__ float red[3][4];
  const int t = blockIdx.x;
  if (t >= T) {
    for (int c = threadIdx.x * 16; c < d; c += 4096) *reinterpret_cast<i32x4 *>(x8 + packed_off((size_t)t, c, d)) = i32x4{0, 0, 0, 0};
    if (threadIdx.x == 0) rowc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float *row = a32 + (size_t)t * d;
  float m = 0.f, m_out = 0.f, s4 = 0.f;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
    const unsigned flags = *reinterpret_cast<const unsigned *>(is_out + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float av = fabsf(v[e]), q = av * av;
      m = fmaxf(m, av);
      if ((flags >> (8 * e)) & 0xFFu) m_out = fmaxf(m_out, av); else s4 = __builtin_fmaf(q, q, s4);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m = fmaxf(m, __shfl_xor(m, off, 64)); m_out = fmaxf(m_out, __shfl_xor(m_out, off, 64)); s4 += __shfl_xor(s4, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = m_out; red[2][threadIdx.x >> 6] = s4; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  m_out = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  s4 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  const float sx = m > 0.f ? m / FP8_MAX : 1.f, inv = 1.f / sx;
  for (int c = threadIdx.x * 16; c < d; c += 4096) {
    i32x4 packed;
#pragma unroll
    for (int q = 0; q < 4; ++q) packed[q] = pack4_fp8(*reinterpret_cast<const f32x4 *>(row + c + 4 * q) * inv);
    *reinterpret_cast<i32x4 *>(x8 + packed_off((size_t)t, c, d)) = packed;
  }
  if (threadIdx.x == 0)
    rowc[t] = f32x4{sx, __builtin_sqrtf(FP8_REL_VAR2 / FP8_ABS_VAR) * m_out * inv, z2 * FP8_REL_VAR2 * __builtin_sqrtf(s4),
                    (*valid & PREP_F8) ? 0.f : 1.f};
}
// per-call column constants of the fp8 band (see quant_x_fp8_kernel): the f32 weights at the batch's outlier dims.
// 8 threads per feature row, 32 rows per workgroup.
__global__ __launch_bounds__(256) void gather_wo_fp8_kernel(const float *__restrict__ W, int N, int d, const int *__restrict__ odims,
                                                           const f32x4 *__restrict__ colbf, f32x4 *__restrict__ colc,
                                                           f32x4 *__restrict__ colc_s) {
  __shared__ int s_dims[MAX_OUT];
  if (threadIdx.x < MAX_OUT) s_dims[threadIdx.x] = odims[threadIdx.x];
  __syncthreads();
  const int n_out = odims[MAX_OUT];
  const int n = blockIdx.x * 32 + (threadIdx.x >> 3);
  float so = 0.f;
  for (int j = threadIdx.x & 7; j < n_out; j += 8) {
    const float w = W[(size_t)n * d + s_dims[j]];
    so = __builtin_fmaf(w, w, so);
  }
  so += __shfl_xor(so, 1, 64); so += __shfl_xor(so, 2, 64); so += __shfl_xor(so, 4, 64);
  if ((threadIdx.x & 7) == 0) {
    const f32x4 st = colbf[n];               // (sw8, |W|_4^2, |W|^2, sw8^2)
    const float ratio = FP8_ABS_VAR / FP8_REL_VAR2;
    const f32x4 cc = {st[0], st[1] + ratio * __builtin_sqrtf((float)d) * st[3], st[2], so * 1.0001f + ratio * (float)n_out * st[3]};
    colc[n] = cc;
    if ((n % SAMPLE_STRIDE) == SAMPLE_OFF) colc_s[n / SAMPLE_STRIDE] = cc;
  }
}

// pointers into a prepared buffer for the operand groups `modes` rebuilds (bit 1: int8 operands, bit 2: without the
// fragment-major copies); stats are rebuilt by every mode
inline RowQuantOut row_quant_out(unsigned char *base, const Prepared &p, int modes, bool i8) {
  RowQuantOut o{};
  o.wstat = reinterpret_cast<f32x4 *>(base + p.off_wstat); o.wstat_s = reinterpret_cast<f32x4 *>(base + p.off_wstat_s);
  o.colbf = reinterpret_cast<f32x4 *>(base + p.off_colbf); o.colbf_s = reinterpret_cast<f32x4 *>(base + p.off_colbf_s);
  o.layout = 1;
  if ((modes & 2) && i8) {
    o.wq = reinterpret_cast<signed char *>(base + p.off_wq); o.wqs = reinterpret_cast<signed char *>(base + p.off_wqs);
    o.wqp = reinterpret_cast<signed char *>(base + p.off_wqp); o.wqsp = reinterpret_cast<signed char *>(base + p.off_wqsp);
    o.ds = reinterpret_cast<float *>(base + p.off_ds);
    o.sdtab = reinterpret_cast<const int *>(base + p.off_sdtab);
    if (!(modes & 4)) {
      o.wqf = reinterpret_cast<signed char *>(base + p.off_wqf); o.wqsf = reinterpret_cast<signed char *>(base + p.off_wqsf);
    }
  }
  return o;
}
// which operand groups a prepare / refresh of `modes` leaves valid (Prepared::valid)
inline unsigned prep_valid_bits(int modes, int N, int d) {
  if ((modes & 8) && i8_shape_ok(N, d)) return ((modes & 1) ? PREP_BF16 : 0u) | PREP_F8;   // (fp8 operands sit where int8 ones would)
  return ((modes & 1) ? PREP_BF16 : 0u) | (((modes & 2) && i8_shape_ok(N, d)) ? (PREP_I8 | ((modes & 4) ? 0u : PREP_FRAG)) : 0u);
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
  // (round 6: eight columns per thread -- 1-KB row pieces, half the workgroups -- measured 0.166 ms of prep against 0.142: reverted)
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
//
// SD (round 6, encode_defs.h "subtractive dither"; the large-batch pass under msae_options::dither): the dims are rounded against
// the SHARED vector r_x(c) of the prepared buffer's seed, q = floor(A + r_x(c)), the pass subtracts the dither again (D_n, prepared
// per feature) and the weights' shared dither through this kernel's per-token integer
//   rowe[t] = (E, m),  E = rint( sum_c dw(c) Aq_c - sum_c dw(c) dx(c) ),  Aq_c = the integer the pass multiplies for dim c.
// An outlier dim keeps a remainder in its OWN column: A = m hi + lo with hi = rint(A / m) EXACT in the outlier tile and
// lo = floor(A - m hi + r_x(c)) in column c (|lo| <= m / 2 + 1 <= 127 for m <= SD_M_EXACT): every dim then carries the same
// one-step uniform residual, the band loses its outlier term (rowc[t][1] = M_t = 0) and |W_n|^2 runs over all dims.  Tokens
// with a larger multiplier keep the coarse outlier steps (hi stochastically rounded per token as before, column c left at 0 -- which
// under the subtraction reads as -dx(c): the same uniform residual): M_t = sqrt(3) m (Hoeffding's 1/4 against the band's 1/12).
template <int SRC, bool FROM_X, bool SD = false>
__global__ __launch_bounds__(256) void quant_x_kernel(const void *__restrict__ x, const float *__restrict__ b_dec, int T, int d,
                                                      const int *__restrict__ odims,
                                                      const unsigned char *__restrict__ is_out,
                                                      signed char *__restrict__ xq,
                                                      signed char *__restrict__ xqo,
                                                      f32x4 *__restrict__ rowc, float zz12, int tile_major,
                                                      const unsigned *__restrict__ valid, unsigned need,
                                                      unsigned long long seed,
                                                      const unsigned long long *__restrict__ dseed_p = nullptr,
                                                      int2 *__restrict__ rowe = nullptr,
                                                      const int *__restrict__ sdtab = nullptr) {
  __shared__ float red[3][4];
  __shared__ long long red64[4];
  __shared__ float red_e0[4];
  const int t = blockIdx.x;
  // msae_options::dither: q = floor(v / step + r(t, c)) instead of rint -- the residual of every dim is zero-mean whatever the
  // token is (dims below one step round to 0 or +-1 at random).  The E0 guard below stays as it is: a token it flags has a
  // scale so coarse that its band separates nothing -- straight to the exact path instead of through a re-score that would
  // flag it anyway (reason 32 / 4); e0 counts the dims that round-to-nearest would zero, whichever rounding runs.
  const bool dith = seed != 0ull;
  const unsigned dkey = dith ? dither_key(seed, (unsigned)t) : 0u;
  auto xq_at = [&](int c) { return xq + (tile_major ? packed_off((size_t)t, c, d, tile_major) : (size_t)t * d + c); };
  if (t >= T) {
    for (int c = threadIdx.x * 16; c < d; c += 4096) *reinterpret_cast<i32x4 *>(xq_at(c)) = i32x4{0, 0, 0, 0};
    if (threadIdx.x < 8) *reinterpret_cast<i32x4 *>(xqo + (size_t)t * MAX_OUT + threadIdx.x * 16) = i32x4{0, 0, 0, 0};
    if (threadIdx.x == 0) { rowc[t] = f32x4{0.f, SD ? 0.f : 1.f, 0.f, 0.f}; if constexpr (SD) rowe[t] = int2{0, 1}; }
    return;
  }
  // (SD: the buffer's seed; 0 = operands rounded to nearest behind a dithering call -- the stale-operand bit below sends the
  // call to the exact path in that case, see run_fast)
  [[maybe_unused]] const unsigned long long dseed = SD ? *dseed_p : 0ull;
  [[maybe_unused]] int e1 = 0;                     // 16 dims' share of sum g_w Aq over the token's own columns (2^-17 units; flushed per iteration)
  [[maybe_unused]] long long e1_64 = 0;
  [[maybe_unused]] long long e_out = 0;            // ... of the outlier tile's entries (g_w hi m)
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
  // the GEMM multiplies the outlier tile's int32 accumulator (|acc| <= 128 * 127 * 127 < 2^21) by m with a 24-bit multiply and
  // keeps the product in int32: m <= M_MAX keeps it below 2^31.  A token whose outlier dims dwarf its inliers by more than that
  // is clamped AND handed to the exact path (guard below): its coarse values would be wrong, not merely noisy.
  constexpr int M_MAX = 1040;
  const bool m_over = m > M_MAX;
  m = m < 1 ? 1 : (m > M_MAX ? M_MAX : m);
  [[maybe_unused]] const bool coarse_out = m > SD_M_EXACT;   // SD: outlier dims without a remainder plane
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
      [[maybe_unused]] i32x4 tb = {0, 0, 0, 0};
      if constexpr (SD) tb = *reinterpret_cast<const i32x4 *>(sdtab + c + 4 * q);
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool outl = ((flags >> (8 * e)) & 0xFFu) != 0;
        const float sv = v[e] * inv;
        int iv;
        if constexpr (SD) {
          const int hx = sd_hx(tb[e]), gw = sd_g(sd_hw(tb[e]));
          // (an outlier dim: what its tile's entry rint(A / m) leaves -- the same float expression as below, so both agree)
          const float rem = outl ? (coarse_out ? 0.f : sv - (float)m * rintf(v[e] * inv_o)) : sv;
          iv = (outl && coarse_out) ? 0 : (int)floorf(rem + sd_r(hx));
          iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
          e1 += gw * iv;
        } else {
          iv = outl ? 0 : (dith ? (int)floorf(sv + dither01(dkey, (unsigned)(c + 4 * q + e))) : (int)rintf(sv));
          iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
        }
        e0 += (!outl && fabsf(sv) <= 0.5f) ? v[e] * v[e] : 0.f;
        w |= ((unsigned)iv & 0xFFu) << (8 * e);
      }
      packed[q] = (int)w;
    }
    *reinterpret_cast<i32x4 *>(xq_at(c)) = packed;
    if constexpr (SD) { e1_64 += e1; e1 = 0; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) e0 += __shfl_xor(e0, off, 64);
  if constexpr (!SD) {
    __syncthreads();                             // red[] of the first reduction has been consumed
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = e0;
    __syncthreads();
    if (threadIdx.x == 0) {
      e0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      // (stale operands, Prepared::valid: the candidate pass would read old weights -- every token to the exact path)
      const float guard = (e0 > GUARD_E0_SX * GUARD_E0_SX * scale * scale || (*valid & need) != need || m_over) ? 1.f : 0.f;
      // (dither: the weights' residuals multiply the DEQUANTISED activations, |A + delta| <= |A| + 1 step per dim (m steps on each
      // of the batch's n_o outlier dims) -- the cross term of the two roundings is inside the Hoeffding proxy, not beside it: the round-5 verdict's item 2)
      const float n_o = (float)odims[MAX_OUT];
      const float an = __builtin_sqrtf(ss) + (dith ? scale * __builtin_sqrtf(((float)d - n_o) + (float)m * (float)m * n_o) : 0.f);
      rowc[t] = f32x4{scale, (float)m, zz12 * an * an, guard};
    }
  }
  if (threadIdx.x < MAX_OUT) {
    const int dim = odims[threadIdx.x];
    float av = 0.f;
    if (dim >= 0) {
      if constexpr (!FROM_X) av = row32[dim];
      else av = load_x1<SRC>(x, (size_t)t * d + dim) - (b_dec ? b_dec[dim] : 0.f);
    }
    // (the outlier dims' own stream of the hash: indices d .. d + MAX_OUT)
    int iv;
    if constexpr (SD) {
      iv = dim >= 0 ? (coarse_out ? (int)floorf(av * inv_o + dither01(dkey, (unsigned)(d + threadIdx.x))) : (int)rintf(av * inv_o)) : 0;
      iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
      if (dim >= 0) e_out = (long long)sd_g(sd_hw(sdtab[dim])) * ((long long)iv * m);   // the tile's share of Aq
    } else {
      iv = dim >= 0 ? (dith ? (int)floorf(av * inv_o + dither01(dkey, (unsigned)(d + threadIdx.x))) : (int)rintf(av * inv_o)) : 0;
      iv = iv > 127 ? 127 : (iv < -127 ? -127 : iv);
    }
    xqo[(size_t)t * MAX_OUT + threadIdx.x] = (signed char)iv;
  }
  if constexpr (SD) {
    long long es = e1_64 + e_out;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) es += __shfl_xor(es, off, 64);
    if ((threadIdx.x & 63) == 0) { red64[threadIdx.x >> 6] = es; red_e0[threadIdx.x >> 6] = e0; }   // (arrays of their own: ONE barrier)
    __syncthreads();
    if (threadIdx.x == 0) {
      e0 = (red_e0[0] + red_e0[1]) + (red_e0[2] + red_e0[3]);
      es = (red64[0] + red64[1]) + (red64[2] + red64[3]);
      const long long f1 = *reinterpret_cast<const long long *>(sdtab + ((d + 1) & ~1));   // sum g_w g_x (2^-34 units)
      const double ev = (double)es * SD_UNIT - (double)f1 * (SD_UNIT * SD_UNIT);
      const float guard = (e0 > GUARD_E0_SX * GUARD_E0_SX * scale * scale || (*valid & need) != need || m_over || dseed == 0ull) ? 1.f : 0.f;
      // P: the weights' residuals multiply the DEQUANTISED activations A + delta, |A + delta| <= |A| + sqrt(d) / 2 -- the cross term
      // of the two roundings is inside the bound
      // (a token with coarse outlier steps: up to m steps of residual on each of the batch's outlier dims)
      const float an = __builtin_sqrtf(ss) + scale * __builtin_sqrtf(0.25f * (float)d + (coarse_out ? (float)m * (float)m * (float)odims[MAX_OUT] : 0.f));
      rowc[t] = f32x4{scale, coarse_out ? 1.7320509f * (float)m : 0.f, zz12 * an * an, guard};
      rowe[t] = int2{(int)__builtin_rint(ev), m};
    }
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
                                                        f32x4 *__restrict__ colc_p, int skip,
                                                        const float *__restrict__ ds = nullptr, float *__restrict__ cds = nullptr,
                                                        float *__restrict__ cds_s = nullptr, float *__restrict__ cds_p = nullptr) {
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
    // subtractive dither (ds != null; encode_defs.h): variance 1/12 per weight rounding (Q = sw^2; the slack rides in the call's
    // z^2 / 12 factors), every dim of the token carries a one-step residual (Si = |W_n|^2 over ALL dims), So serves the tokens
    // whose outlier multiplier is too large for a remainder plane -- and the row's correction Ds_n in the launch's column orders
    const f32x4 cc = ds ? f32x4{st[0], st[0] * st[0], st[2], so} : f32x4{st[0], st[1], fmaxf(st[2] - so, 0.f), so};
    colc[n] = cc;
    if (samp) colc_s[n / SAMPLE_STRIDE] = cc;
    else if (skip) colc_p[main_row(n)] = cc;
    if (ds) {
      const float dv = ds[n];
      cds[n] = dv;
      if (samp) cds_s[n / SAMPLE_STRIDE] = dv;
      else if (skip) cds_p[main_row(n)] = dv;
    }
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

}  // namespace
