// train.hip -- the parameter-sized passes of one SAE optimisation step, fused.
//
// The reference trainer (train/sae/sae/trainer.py:347-414) runs, per step and per [N, d] matrix,
//   set_decoder_norm_to_unit_norm          sae.py:249-255   norm, +eps, div            3 passes
//   clip_grad_norm_(params, 1.0)           trainer.py:390   norm over all grads, g *= c 3 passes
//   remove_gradient_parallel_to_decoder... sae.py:257-271   (g*W).sum, g -= along*W     ~9 passes
//   Adam                                   trainer.py:395   p, g, m, v -> p, m, v       7 passes
// as separate torch ops: ~67 GB of HBM traffic at d=4096, N=131072.  Here:
//   unit_norm_rows   one read + one write of W_dec (the row is re-read from L2 for the scale)
//   grad_sumsq       one read of each gradient -> a device scalar (no host sync)
//   adam_rows        one pass: clip coefficient from the device scalar, optional removal of the
//                    row-parallel component, Adam moments and parameter update.  g is not written.
// ~38 GB.  All HBM-bound streaming kernels: 16-B accesses, one workgroup per row so the row's
// second touch (projection / scale) is an L2 hit.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();                 // red[] may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// W[r][:] /= (||W[r][:]||_2 + eps)      (sae.py:252-255)
__global__ __launch_bounds__(256) void unit_norm_rows_kernel(float *__restrict__ W, int d, float eps) {
  __shared__ float red[4];
  float *row = W + (size_t)blockIdx.x * d;
  float ss = 0.f;
  if ((d & 3) == 0) {
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
  } else {
    for (int c = threadIdx.x; c < d; c += 256) ss += row[c] * row[c];
  }
  const float inv = 1.f / (sqrtf(block_sum(ss, red)) + eps);
  if ((d & 3) == 0) {
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
      v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
      *reinterpret_cast<f32x4 *>(row + c) = v;
    }
  } else {
    for (int c = threadIdx.x; c < d; c += 256) row[c] *= inv;
  }
}

// *accum += sum(g[i]^2)
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ g, size_t n,
                                                         float *__restrict__ accum) {
  __shared__ float red[4];
  float ss = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4 *>(g)[i];
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) ss += g[i] * g[i];
  const float s = block_sum(ss, red);
  if (threadIdx.x == 0) atomicAdd(accum, s);
}

struct AdamArgs {
  float lr, beta1, beta2, eps, max_norm;
  float bc1, bc2_sqrt;     // 1 - beta1^t, sqrt(1 - beta2^t)
  int project;
};

// torch's fused Adam arithmetic (aten/native/cuda/fused_adam_utils.cuh, no weight decay / amsgrad)
__device__ __forceinline__ void adam_update(float &w, float g, float &m, float &v, const AdamArgs &a) {
  m = m + (g - m) * (1.f - a.beta1);                 // lerp(m, g, 1 - beta1)
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float step_size = a.lr / a.bc1;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  w -= step_size * (m / denom);                      // addcdiv_(m, denom, -step_size)
}

// One workgroup per row r of W/G/M/V [rows][d]:
//   c     = min(1, max_norm / (sqrt(*total_sumsq) + 1e-6))        (clip_grad_norm_)
//   g     = c * G[r][:]
//   g    -= <g, W[r][:]> * W[r][:]      if project                (sae.py:266-271, after clipping)
//   Adam(W, g, M, V)
__global__ __launch_bounds__(256) void adam_rows_kernel(float *__restrict__ W, const float *__restrict__ G,
                                                        float *__restrict__ M, float *__restrict__ V,
                                                        int d, const float *__restrict__ total_sumsq,
                                                        AdamArgs a) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * d;
  float clip = 1.f;
  if (total_sumsq) {
    const float c = a.max_norm / (sqrtf(*total_sumsq) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  float along = 0.f;
  if (a.project) {
    float dot = 0.f;
    if ((d & 3) == 0) {
      for (int c = threadIdx.x * 4; c < d; c += 1024) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(G + base + c);
        const f32x4 w = *reinterpret_cast<const f32x4 *>(W + base + c);
        dot += (g[0] * clip) * w[0] + (g[1] * clip) * w[1] + (g[2] * clip) * w[2] + (g[3] * clip) * w[3];
      }
    } else {
      for (int c = threadIdx.x; c < d; c += 256) dot += (G[base + c] * clip) * W[base + c];
    }
    along = block_sum(dot, red);
  }
  if ((d & 3) == 0) {
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
      const f32x4 g = *reinterpret_cast<const f32x4 *>(G + base + c);     // L2 hit when projecting
      f32x4 w = *reinterpret_cast<const f32x4 *>(W + base + c);
      f32x4 m = *reinterpret_cast<const f32x4 *>(M + base + c);
      f32x4 v = *reinterpret_cast<const f32x4 *>(V + base + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ge = g[e] * clip, we = w[e], me = m[e], ve = v[e];
        if (a.project) ge -= along * we;
        adam_update(we, ge, me, ve, a);
        w[e] = we; m[e] = me; v[e] = ve;
      }
      *reinterpret_cast<f32x4 *>(W + base + c) = w;
      *reinterpret_cast<f32x4 *>(M + base + c) = m;
      *reinterpret_cast<f32x4 *>(V + base + c) = v;
    }
  } else {
    for (int c = threadIdx.x; c < d; c += 256) {
      float ge = G[base + c] * clip, w = W[base + c], m = M[base + c], v = V[base + c];
      if (a.project) ge -= along * w;
      adam_update(w, ge, m, v, a);
      W[base + c] = w; M[base + c] = m; V[base + c] = v;
    }
  }
}

}  // namespace

extern "C" int msae_unit_norm_rows_f32(float *W, int N, int d, float eps, void *stream) {
  if (!W || N <= 0 || d <= 0) return MSAE_EINVAL;
  if ((d & 3) == 0 && !msae_aligned(W, 16)) return MSAE_EALIGN;
  hipLaunchKernelGGL(unit_norm_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, d, eps);
  return msae_launch_status();
}

extern "C" int msae_grad_sumsq_f32(const float *g, size_t n, float *accum, void *stream) {
  if (!g || !accum) return MSAE_EINVAL;
  if (!msae_aligned(g, 16)) return MSAE_EALIGN;
  if (n == 0) return 0;
  const size_t want = (n / 4 + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, accum);
  return msae_launch_status();
}

extern "C" int msae_adam_rows_f32(float *W, const float *G, float *M, float *V, int rows, int d,
                                  const float *total_sumsq, float max_norm, int project, float lr,
                                  float beta1, float beta2, float eps, int step, void *stream) {
  if (!W || !G || !M || !V || rows <= 0 || d <= 0 || step < 1) return MSAE_EINVAL;
  if ((d & 3) == 0 && !(msae_aligned(W, 16) && msae_aligned(G, 16) && msae_aligned(M, 16) && msae_aligned(V, 16)))
    return MSAE_EALIGN;
  AdamArgs a{};
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.project = project ? 1 : 0;
  hipLaunchKernelGGL(adam_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, W, G, M, V, d,
                     total_sumsq, a);
  return msae_launch_status();
}
