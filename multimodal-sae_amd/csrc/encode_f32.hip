// encode_f32.hip -- exact f32 encoder GEMM: out[T][N] = relu((x - b_dec) W_enc^T + b_enc).
//
// Replaces Sae.pre_acts (reference sae/sae.py:172-177: nn.Linear in f32 + ReLU).
// Roofline: f32 MFMA (157 TFLOP/s dense on gfx950; there is no TF32/xf32).  2*d*N FLOP per token.
//
// v_mfma_f32_32x32x2_f32 computes, per output element, fma(a_k1, b_k1, fma(a_k0, b_k0, c)) with
// lanes 0-31 carrying k0 and lanes 32-63 carrying k1.  The K loop walks k in ascending order
// with ONE accumulator per output, so every pre-activation is the same ascending-k f32 fma chain
// as oracle/sae_oracle.c:msae_oracle_pre_acts -- compared bit-exactly in tests/.
//
// Tiling: 128 (tokens) x 128 (features) x 32 (k) per workgroup, 4 waves as 2x2, each wave a
// 64x64 tile = 2x2 MFMA blocks (64 accumulator VGPRs).  Operands are staged global -> registers
// -> LDS (x is up-cast and b_dec subtracted on the way), double-buffered, one barrier per k-tile.
// LDS rows are padded to 33 floats: fragment reads (one f32 per lane, 32 rows x same k) and
// staging writes are bank-conflict-free.  MFMA issue is the bound: 4 MFMAs (256 cycles/SIMD) per
// 4 ds_read_b32.
#include "common.h"

namespace {

constexpr int F_BM = 128, F_BN = 128, F_BK = 32, F_PITCH = 33, F_THREADS = 256;
constexpr int F_LDS_FLOATS = 2 * (F_BM + F_BN) * F_PITCH;  // double-buffered A and B tiles

struct StageRegs {
  f32x4 a[4];
  f32x4 b[4];
};

// Optional row gather: when `rows` != NULL token t of the tile reads x[rows[t]] and the tile
// count comes from *n_rows (device side), so the exact path can re-compute a compacted list of
// tokens flagged by the fused encoder without a host round trip.
template <int DT, bool VEC>
__device__ __forceinline__ void stage_load(StageRegs &r, const void *x, const float *W,
                                           const float *b_dec, const int *rows, int T, int d, int N,
                                           int m0, int n0, int k0) {
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
  const int kq = k0 + q * 4;
  f32x4 bd = {0.f, 0.f, 0.f, 0.f};
  if (b_dec) {
    if constexpr (VEC) {
      if (kq < d) bd = *reinterpret_cast<const f32x4 *>(b_dec + kq);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bd[e] = (kq + e < d) ? b_dec[kq + e] : 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
    const int t = m0 + row;
    if (t < T) {
      const size_t xr = rows ? (size_t)rows[t] : (size_t)t;
      if constexpr (VEC) {
        if (kq < d) av = load_x4<DT>(x, xr * d + kq) - bd;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kq + e < d) av[e] = load_x1<DT>(x, xr * d + kq + e) - bd[e];
      }
    }
    const int n = n0 + row;
    if (n < N) {
      if constexpr (VEC) {
        if (kq < d) bv = *reinterpret_cast<const f32x4 *>(W + (size_t)n * d + kq);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kq + e < d) bv[e] = W[(size_t)n * d + kq + e];
      }
    }
    r.a[it] = av;
    r.b[it] = bv;
  }
}

__device__ __forceinline__ void stage_store(const StageRegs &r, float *sA, float *sB) {
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    float *pa = sA + row * F_PITCH + q * 4;
    float *pb = sB + row * F_PITCH + q * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[e] = r.a[it][e];
      pb[e] = r.b[it][e];
    }
  }
}

template <int DT, bool VEC>
__global__ __launch_bounds__(F_THREADS, 2) void pre_acts_f32_kernel(
    const void *__restrict__ x, const float *__restrict__ W, const float *__restrict__ b_enc,
    const float *__restrict__ b_dec, const int *__restrict__ rows, const int *__restrict__ n_rows,
    int T, int d, int N, int relu, float *__restrict__ out, int ld_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (n_rows) T = min(T, *n_rows);
  constexpr int STAGE = (F_BM + F_BN) * F_PITCH;  // floats per stage: A tile then B tile
  // row tiles blockIdx.y, +gridDim.y, ... and column tiles blockIdx.x, +gridDim.x, ...: with a device-side row count the
  // launch is sized for a few tiles only and a workgroup walks as many as the count needs (none -> it leaves at once: the
  // empty launch of the in-call exact fallback is 512 workgroups, not N / 128 * 2)
  for (int n0 = blockIdx.x * F_BN; n0 < N; n0 += gridDim.x * F_BN)
  for (int m0 = blockIdx.y * F_BM; m0 < T; m0 += gridDim.y * F_BM) {

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, khalf = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (d + F_BK - 1) / F_BK;
  StageRegs regs;
  stage_load<DT, VEC>(regs, x, W, b_dec, rows, T, d, N, m0, n0, 0);
  stage_store(regs, smem, smem + F_BM * F_PITCH);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage_load<DT, VEC>(regs, x, W, b_dec, rows, T, d, N, m0, n0, (kt + 1) * F_BK);
    const float *a_base = smem + cur * STAGE + (wr * 64 + l31) * F_PITCH + khalf;
    const float *b_base = smem + cur * STAGE + F_BM * F_PITCH + (wc * 64 + l31) * F_PITCH + khalf;
#pragma unroll
    for (int s = 0; s < F_BK / 2; ++s) {
      const float a0 = a_base[2 * s], a1 = a_base[32 * F_PITCH + 2 * s];
      const float b0 = b_base[2 * s], b1 = b_base[32 * F_PITCH + 2 * s];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) stage_store(regs, smem + (cur ^ 1) * STAGE, smem + (cur ^ 1) * STAGE + F_BM * F_PITCH);
    __syncthreads();
  }

  // epilogue: C[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + l31;
    if (n >= N) continue;
    const float bn = b_enc ? b_enc[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
        if (t < T) {
          float v = acc[i][j][e] + bn;
          if (relu && !(v > 0.f)) v = 0.f;
          out[(size_t)t * ld_out + n] = v;
        }
      }
    }
  }
  __syncthreads();   // LDS stages are rewritten by the next row tile
  }
}

template <int DT>
int launch_dt(const void *x, const float *W, const float *b_enc, const float *b_dec, const int *rows,
              const int *n_rows, int T, int d, int N, int relu, float *out, int ld_out,
              hipStream_t s) {
  const size_t xb = (DT == MSAE_F32) ? 16 : 8;
  const bool vec = (d % 4 == 0) && msae_aligned(x, xb) && msae_aligned(W, 16) &&
                   (!b_dec || msae_aligned(b_dec, 16));
  int tiles_m = (T + F_BM - 1) / F_BM;
  int tiles_n = (N + F_BN - 1) / F_BN;
  if (n_rows && tiles_m > 2) tiles_m = 2;   // device-side count: workgroups loop over the row tiles ...
  if (n_rows && tiles_n > 256) tiles_n = 256;   // ... and over the column tiles
  dim3 grid(tiles_n, tiles_m);
  const size_t smem = F_LDS_FLOATS * sizeof(float);
  if (vec) {
    auto kern = pre_acts_f32_kernel<DT, true>;
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
    hipLaunchKernelGGL(kern, grid, dim3(F_THREADS), smem, s, x, W, b_enc, b_dec, rows, n_rows, T, d,
                       N, relu, out, ld_out);
  } else {
    auto kern = pre_acts_f32_kernel<DT, false>;
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
    hipLaunchKernelGGL(kern, grid, dim3(F_THREADS), smem, s, x, W, b_enc, b_dec, rows, n_rows, T, d,
                       N, relu, out, ld_out);
  }
  return msae_launch_status();
}

}  // namespace

// Shared with encode_fused.hip (exact recompute of flagged tokens through a row list).
int msae_pre_acts_launch(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                         const float *b_dec, const int *rows, const int *n_rows, int T, int d, int N,
                         int relu, float *out, int ld_out, hipStream_t s) {
  if (T < 0 || d <= 0 || N <= 0 || ld_out < N) return MSAE_EINVAL;
  if (T == 0) return 0;
  if (!n_rows && (T + F_BM - 1) / F_BM > 65535) return MSAE_ENOTIMPL;
  switch (x_dtype) {
    case MSAE_F32: return launch_dt<MSAE_F32>(x, W_enc, b_enc, b_dec, rows, n_rows, T, d, N, relu, out, ld_out, s);
    case MSAE_BF16: return launch_dt<MSAE_BF16>(x, W_enc, b_enc, b_dec, rows, n_rows, T, d, N, relu, out, ld_out, s);
    case MSAE_F16: return launch_dt<MSAE_F16>(x, W_enc, b_enc, b_dec, rows, n_rows, T, d, N, relu, out, ld_out, s);
    default: return MSAE_EINVAL;
  }
}

extern "C" int msae_pre_acts_f32(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                 const float *b_dec, int T, int d, int N, int relu, float *out,
                                 void *stream) {
  return msae_pre_acts_launch(x, x_dtype, W_enc, b_enc, b_dec, nullptr, nullptr, T, d, N, relu, out,
                              N, (hipStream_t)stream);
}
