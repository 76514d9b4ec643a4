// decode.hip -- k-sparse decoder (gather matmul over W_dec rows) and its backward pieces.
//
// Replaces TritonDecoder / triton_sparse_dense_matmul (reference sae/kernels.py:178-284,403-429).
// Roofline: HBM / L2 bandwidth.  Algorithmic bytes per token = k*d*4 (gathered rows) + k*8
// (idx, acts) + d*4 (output) [+ d*4 b_dec, L2-resident]; 0.5 FLOP/B.
//
// Layout: one workgroup column-slab of 1024 floats per token; lane i owns one float4 of the
// output row, so each gathered W_dec row segment is read as 64 lanes x 16 B = 1 KiB coalesced
// per wave instruction.  (idx, acts) are wave-uniform scalar loads.  The j-loop is unrolled so 8
// independent 16-B loads per lane are in flight; the f32 fma chain runs in j order so the result
// is bit-identical to oracle/sae_oracle.c:msae_oracle_decode.
#include "common.h"

namespace {

constexpr int DEC_THREADS = 256;
constexpr int DEC_UNROLL = 8;

// grid: (A, ceil(d / 1024))
__global__ __launch_bounds__(DEC_THREADS) void decode_fwd_v4_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ acts,
    const float *__restrict__ W_dec, const float *__restrict__ b_dec, float *__restrict__ out,
    int k, int N, int d, int32_t *status) {
  const int a = blockIdx.x;
  const int col = (blockIdx.y * DEC_THREADS + threadIdx.x) * 4;
  const int32_t *ip = idx + (size_t)a * k;
  const float *vp = acts + (size_t)a * k;
  if (col >= d) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int j = 0;
  for (; j + DEC_UNROLL <= k; j += DEC_UNROLL) {
    f32x4 w[DEC_UNROLL];
    float v[DEC_UNROLL];
#pragma unroll
    for (int u = 0; u < DEC_UNROLL; ++u) {
      int i = ip[j + u];
      v[u] = vp[j + u];
      const bool bad = (unsigned)i >= (unsigned)N;
      if (bad) {
        v[u] = 0.f;
        i = 0;
        if (status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(status, 1);
      }
      w[u] = *reinterpret_cast<const f32x4 *>(W_dec + (size_t)i * d + col);
    }
#pragma unroll
    for (int u = 0; u < DEC_UNROLL; ++u) {
      f32x4 f;
      f[0] = __builtin_fmaf(v[u], w[u][0], acc[0]);
      f[1] = __builtin_fmaf(v[u], w[u][1], acc[1]);
      f[2] = __builtin_fmaf(v[u], w[u][2], acc[2]);
      f[3] = __builtin_fmaf(v[u], w[u][3], acc[3]);
      acc = (v[u] == 0.f) ? acc : f;  // kernels.py:277: zero activations contribute nothing
    }
  }
  for (; j < k; ++j) {
    int i = ip[j];
    float v = vp[j];
    if ((unsigned)i >= (unsigned)N) {
      v = 0.f;
      i = 0;
      if (status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(status, 1);
    }
    f32x4 w = *reinterpret_cast<const f32x4 *>(W_dec + (size_t)i * d + col);
    f32x4 f;
    f[0] = __builtin_fmaf(v, w[0], acc[0]);
    f[1] = __builtin_fmaf(v, w[1], acc[1]);
    f[2] = __builtin_fmaf(v, w[2], acc[2]);
    f[3] = __builtin_fmaf(v, w[3], acc[3]);
    acc = (v == 0.f) ? acc : f;
  }
  if (b_dec) {
    f32x4 b = *reinterpret_cast<const f32x4 *>(b_dec + col);
    acc[0] += b[0]; acc[1] += b[1]; acc[2] += b[2]; acc[3] += b[3];
  }
  *reinterpret_cast<f32x4 *>(out + (size_t)a * d + col) = acc;
}

// any d / alignment: one column per lane.  grid: (A, ceil(d / 256))
__global__ __launch_bounds__(DEC_THREADS) void decode_fwd_scalar_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ acts,
    const float *__restrict__ W_dec, const float *__restrict__ b_dec, float *__restrict__ out,
    int k, int N, int d, int32_t *status) {
  const int a = blockIdx.x;
  const int col = blockIdx.y * DEC_THREADS + threadIdx.x;
  if (col >= d) return;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    int i = idx[(size_t)a * k + j];
    float v = acts[(size_t)a * k + j];
    if ((unsigned)i >= (unsigned)N) {
      v = 0.f;
      i = 0;
      if (status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(status, 1);
    }
    float f = __builtin_fmaf(v, W_dec[(size_t)i * d + col], acc);
    acc = (v == 0.f) ? acc : f;
  }
  if (b_dec) acc += b_dec[col];
  out[(size_t)a * d + col] = acc;
}

// g_acts[a][j] = grad_out[a][:] . W_dec[idx[a][j]][:]      (kernels.py:341-400)
// one wave per (a, j); lanes stride the row in float4s, then a wave reduction.
__global__ __launch_bounds__(256) void decode_bwd_acts_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ grad_out,
    const float *__restrict__ W_dec, int A, int k, int N, int d, float *__restrict__ g_acts) {
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)A * k) return;
  const int a = (int)(pair / k);
  int i = idx[pair];
  if ((unsigned)i >= (unsigned)N) i = 0;
  const float *g = grad_out + (size_t)a * d;
  const float *w = W_dec + (size_t)i * d;
  float acc = 0.f;
  if ((d & 3) == 0) {
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 gv = *reinterpret_cast<const f32x4 *>(g + c);
      f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);
      acc = __builtin_fmaf(gv[0], wv[0], acc);
      acc = __builtin_fmaf(gv[1], wv[1], acc);
      acc = __builtin_fmaf(gv[2], wv[2], acc);
      acc = __builtin_fmaf(gv[3], wv[3], acc);
    }
  } else {
    for (int c = lane; c < d; c += 64) acc = __builtin_fmaf(g[c], w[c], acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) g_acts[pair] = acc;
}

// g_W_dec[idx[a][j]][:] += acts[a][j] * grad_out[a][:]      (kernels.py:10-175)
// grid: (A*k, ceil(d/256)); hardware f32 atomics (rows touched by several tokens collide).
__global__ __launch_bounds__(256) void decode_bwd_wdec_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ acts,
    const float *__restrict__ grad_out, int k, int N, int d, float *__restrict__ g_W) {
  const long pair = blockIdx.x;
  const int a = (int)(pair / k);
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= d) return;
  const int i = idx[pair];
  const float v = acts[pair];
  if ((unsigned)i >= (unsigned)N || v == 0.f) return;
  unsafeAtomicAdd(g_W + (size_t)i * d + col, v * grad_out[(size_t)a * d + col]);
}

}  // namespace

extern "C" int msae_decode_f32(const int32_t *idx, const float *acts, const float *W_dec,
                               const float *b_dec, int A, int k, int N, int d, float *out,
                               int32_t *status, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0) return MSAE_EINVAL;
  if (A == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (d % 4 == 0) && msae_aligned(W_dec, 16) && msae_aligned(out, 16) &&
                   (!b_dec || msae_aligned(b_dec, 16));
  if (vec) {
    dim3 grid(A, (d + DEC_THREADS * 4 - 1) / (DEC_THREADS * 4));
    hipLaunchKernelGGL(decode_fwd_v4_kernel, grid, dim3(DEC_THREADS), 0, s, idx, acts, W_dec, b_dec,
                       out, k, N, d, status);
  } else {
    dim3 grid(A, (d + DEC_THREADS - 1) / DEC_THREADS);
    hipLaunchKernelGGL(decode_fwd_scalar_kernel, grid, dim3(DEC_THREADS), 0, s, idx, acts, W_dec,
                       b_dec, out, k, N, d, status);
  }
  return msae_launch_status();
}

extern "C" int msae_decode_bwd_acts_f32(const int32_t *idx, const float *grad_out,
                                        const float *W_dec, int A, int k, int N, int d,
                                        float *g_acts, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0) return MSAE_EINVAL;
  if (A == 0) return 0;
  if (d % 4 == 0 && (!msae_aligned(W_dec, 16) || !msae_aligned(grad_out, 16))) return MSAE_EALIGN;
  const long pairs = (long)A * k;
  hipLaunchKernelGGL(decode_bwd_acts_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, idx, grad_out, W_dec, A, k, N, d, g_acts);
  return msae_launch_status();
}

extern "C" int msae_decode_bwd_wdec_f32(const int32_t *idx, const float *acts,
                                        const float *grad_out, int A, int k, int N, int d,
                                        float *g_W_dec, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0) return MSAE_EINVAL;
  if (A == 0) return 0;
  dim3 grid((unsigned)((long)A * k), (d + 255) / 256);
  hipLaunchKernelGGL(decode_bwd_wdec_kernel, grid, dim3(256), 0, (hipStream_t)stream, idx, acts,
                     grad_out, k, N, d, g_W_dec);
  return msae_launch_status();
}
