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
#include "encode_defs.h"
#include "encode_prep.h"

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

// *accum += sum(v[0 .. n)) in a FIXED order (one workgroup: per-thread strided partial sums, then block_sum): the total of
// the per-row squared gradient norms the weight-gradient kernel wrote (msae_decode_bwd_wdec_f32: row_sumsq)
__global__ __launch_bounds__(1024) void sum_fixed_kernel(const float *__restrict__ v, size_t n, float *__restrict__ accum) {
  __shared__ float red[16];
  float ss = 0.f;
  const size_t n4 = n / 4;                               // (v comes from the caching allocator: 16-B aligned)
  for (size_t i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 q = reinterpret_cast<const f32x4 *>(v)[i];
    ss += (q[0] + q[1]) + (q[2] + q[3]);
  }
  for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) ss += v[i];
  const float t = block_sum(ss, red);
  if (threadIdx.x == 0) *accum += t;
}

// out[c] = scale * sum_n s[n] W[n][c]: the b_dec gradient of the sparse encoder backward, -sum_t sum_j g[t,j] W_enc[n_tj] =
// -(s^T W_enc) with s[n] = the summed latent gradients of feature n (msae_decode_bwd_wdec_f32: row_act_sum) -- ONE streaming read
// of W_enc instead of a k-row gather per token plus a sum over tokens.  Stage 1: workgroup b sums its 64 consecutive rows (ascending)
// into part[b][:]; stage 2 adds the partial rows up in a fixed order.  Bit-reproducible.
constexpr int WRS_ROWS = 64;       // rows per workgroup of stage 1
__global__ __launch_bounds__(256) void weighted_rows_part_kernel(const float *__restrict__ W, const float *__restrict__ sv, int N,
                                                                 int d, float *__restrict__ part) {
  __shared__ float s_w[WRS_ROWS];
  const int n0 = blockIdx.x * WRS_ROWS;
  if (threadIdx.x < WRS_ROWS) s_w[threadIdx.x] = n0 + (int)threadIdx.x < N ? sv[n0 + threadIdx.x] : 0.f;
  __syncthreads();
  const int rows = min(WRS_ROWS, N - n0);
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *base = W + (size_t)n0 * d + c;
    for (int r0 = 0; r0 < rows; r0 += 8) {            // eight rows' loads in flight per thread; rows with s == 0 are not read
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool live = r0 + u < rows && s_w[r0 + u] != 0.f;           // workgroup-uniform
        v[u] = live ? *reinterpret_cast<const f32x4 *>(base + (size_t)(r0 + u) * d) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {                    // ascending row order: a fixed summation order
        const float w = r0 + u < rows ? s_w[r0 + u] : 0.f;
        acc[0] = __builtin_fmaf(w, v[u][0], acc[0]); acc[1] = __builtin_fmaf(w, v[u][1], acc[1]);
        acc[2] = __builtin_fmaf(w, v[u][2], acc[2]); acc[3] = __builtin_fmaf(w, v[u][3], acc[3]);
      }
    }
    *reinterpret_cast<f32x4 *>(part + (size_t)blockIdx.x * d + c) = acc;
  }
}
// out[y][c] = scale * sum of part rows [y per, (y + 1) per) (blockIdx.y = y): run twice -- 2048 partial rows -> 64 -> 1 -- so that
// no thread walks more than 64 rows (one 16-block launch over 2048 rows took 0.22 ms)
__global__ __launch_bounds__(256) void weighted_rows_sum_kernel(const float *__restrict__ part, int groups, int per, int d,
                                                                float scale, float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int b0 = blockIdx.y * per, b1 = min(groups, b0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;       // four interleaved partial sums (fixed assignment), then a fixed combine
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    a0 += part[(size_t)b * d + c]; a1 += part[(size_t)(b + 1) * d + c];
    a2 += part[(size_t)(b + 2) * d + c]; a3 += part[(size_t)(b + 3) * d + c];
  }
  for (; b < b1; ++b) a0 += part[(size_t)b * d + c];
  out[(size_t)blockIdx.y * d + c] = scale * ((a0 + a1) + (a2 + a3));
}

// adam_rows_kernel with the NEXT step's passes over the same matrix folded in (d % 4 == 0; the updated row stays in registers
// for d <= 8192):
//   renorm_eps >= 0   W[r] /= |W[r]| + eps after the update: the trainer's set_decoder_norm_to_unit_norm at the top of the next
//                     step (sae.py:249-255, trainer.py:352) -- same arithmetic, reduction order and bits as unit_norm_rows_kernel,
//                     without its read + write of the 2 GiB matrix
//   REFRESH 1 / 2     the coarse-pass operands of the updated encoder row (int8: row statistics + quantised copies, as
//                     row_stats_quant_kernel<true>; bf16: statistics + the bf16 copy) into the prepared buffer -- what
//                     msae_encoder_refresh_for would rebuild with another sweep over W_enc before the next encode
struct FusedTail {
  float renorm_eps;
  RowQuantOut rq;
  unsigned short *wb, *ws;      // REFRESH 2: bf16 copy + bf16 sample rows
};
template <int REFRESH>
__global__ __launch_bounds__(256) void adam_rows_fused_kernel(float *__restrict__ W, const float *__restrict__ G,
                                                              float *__restrict__ M, float *__restrict__ V,
                                                              int d, const float *__restrict__ total_sumsq,
                                                              AdamArgs a, FusedTail ft) {
  __shared__ float red[4];
  __shared__ float red3[3][4];
  const size_t base = (size_t)blockIdx.x * d;
  float clip = 1.f;
  if (total_sumsq) {
    const float c = a.max_norm / (sqrtf(*total_sumsq) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  float along = 0.f;
  if (a.project) {
    float dot = 0.f;
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
      const f32x4 g = *reinterpret_cast<const f32x4 *>(G + base + c);
      const f32x4 w = *reinterpret_cast<const f32x4 *>(W + base + c);
      dot += (g[0] * clip) * w[0] + (g[1] * clip) * w[1] + (g[2] * clip) * w[2] + (g[3] * clip) * w[3];
    }
    along = block_sum(dot, red);
  }
  const bool renorm = ft.renorm_eps >= 0.f;
  constexpr int KEEP = 8;                         // d <= 8192 (checked by the host)
  f32x4 keep[KEEP];
  float ss = 0.f;
  {
    int it = 0;
    for (int c = threadIdx.x * 4; c < d; c += 1024, ++it) {
      // the moments are touched exactly once per step (and the gradient, unless the projection has just read it): streaming
      // accesses, kept out of the caches' way (MSAE_ADAM_PLAIN_LOADS in a tuning build switches the hint off)
      const f32x4 g = a.project ? *reinterpret_cast<const f32x4 *>(G + base + c)     // L2 hit when projecting
                                : MSAE_ADAM_LOAD(reinterpret_cast<const f32x4 *>(G + base + c));
      f32x4 w = *reinterpret_cast<const f32x4 *>(W + base + c);
      f32x4 m = MSAE_ADAM_LOAD(reinterpret_cast<const f32x4 *>(M + base + c));
      f32x4 v = MSAE_ADAM_LOAD(reinterpret_cast<const f32x4 *>(V + base + c));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ge = g[e] * clip, we = w[e], me = m[e], ve = v[e];
        if (a.project) ge -= along * we;
        adam_update(we, ge, me, ve, a);
        w[e] = we; m[e] = me; v[e] = ve;
      }
      MSAE_ADAM_STORE(m, reinterpret_cast<f32x4 *>(M + base + c));
      MSAE_ADAM_STORE(v, reinterpret_cast<f32x4 *>(V + base + c));
      if (renorm) {
#pragma unroll
        for (int q = 0; q < KEEP; ++q) if (q == it) keep[q] = w;
        ss += w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
      } else {
        *reinterpret_cast<f32x4 *>(W + base + c) = w;
      }
    }
  }
  if (renorm) {
    const float inv = 1.f / (sqrtf(block_sum(ss, red)) + ft.renorm_eps);
    int it = 0;
    for (int c = threadIdx.x * 4; c < d; c += 1024, ++it) {
      f32x4 v = keep[0];
#pragma unroll
      for (int q = 1; q < KEEP; ++q) if (q == it) v = keep[q];
      v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
      *reinterpret_cast<f32x4 *>(W + base + c) = v;
    }
  }
  if constexpr (REFRESH != 0) {
    __threadfence_block();
    __syncthreads();                              // the whole updated row is visible to the workgroup (same CU, same L1)
    const int n = blockIdx.x;
    if constexpr (REFRESH == 1) {
      row_stats_quant_row<true>(W, n, d, ft.rq, red3);
    } else {
      row_stats_quant_row<false>(W, n, d, ft.rq, red3);
      const bool samp = (n % SAMPLE_STRIDE) == SAMPLE_OFF;
      for (int c = threadIdx.x * 4; c < d; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(W + base + c);
        u16x4 o;
        o[0] = f32_to_bf16_bits(v[0]); o[1] = f32_to_bf16_bits(v[1]); o[2] = f32_to_bf16_bits(v[2]); o[3] = f32_to_bf16_bits(v[3]);
        *reinterpret_cast<u16x4 *>(ft.wb + base + c) = o;
        if (samp) *reinterpret_cast<u16x4 *>(ft.ws + (size_t)(n / SAMPLE_STRIDE) * d + c) = o;
      }
    }
  }
}

}  // namespace

constexpr int WRS_MID = 64;        // partial rows after the first summation level
extern "C" size_t msae_weighted_row_sum_ws_bytes(int N, int d) {
  return (N > 0 && d > 0) ? (size_t)((N + WRS_ROWS - 1) / WRS_ROWS + WRS_MID) * d * 4 : 0;
}

extern "C" int msae_weighted_row_sum_f32(const float *W, const float *s, int N, int d, float scale, float *out, void *ws,
                                         size_t ws_bytes, void *stream) {
  if (!W || !s || !out || N <= 0 || d <= 0 || (d & 3) != 0) return MSAE_EINVAL;
  if (!ws || ws_bytes < msae_weighted_row_sum_ws_bytes(N, d)) return MSAE_EWS;
  if (!msae_aligned(W, 16) || !msae_aligned(ws, 16)) return MSAE_EALIGN;
  float *part = static_cast<float *>(ws);
  const int groups = (N + WRS_ROWS - 1) / WRS_ROWS;
  hipLaunchKernelGGL(weighted_rows_part_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, W, s, N, d, part);
  float *mid = part + (size_t)groups * d;
  const int per = (groups + WRS_MID - 1) / WRS_MID, n_mid = (groups + per - 1) / per;
  hipLaunchKernelGGL(weighted_rows_sum_kernel, dim3((d + 255) / 256, n_mid), dim3(256), 0, (hipStream_t)stream, part, groups, per, d,
                     1.f, mid);
  hipLaunchKernelGGL(weighted_rows_sum_kernel, dim3((d + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, mid, n_mid, n_mid, d,
                     scale, out);
  return msae_launch_status();
}

extern "C" int msae_sum_f32(const float *v, size_t n, float *accum, void *stream) {
  if (!v || !accum) return MSAE_EINVAL;
  if (!msae_aligned(v, 16)) return MSAE_EALIGN;
  if (n == 0) return 0;
  hipLaunchKernelGGL(sum_fixed_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, accum);
  return msae_launch_status();
}

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

extern "C" int msae_adam_rows_fused_f32(float *W, const float *G, float *M, float *V, int rows, int d,
                                        const float *total_sumsq, float max_norm, int project, float lr,
                                        float beta1, float beta2, float eps, int step, float renorm_eps,
                                        void *prepared, int T_next, const msae_options *opts, void *stream) {
  if (!W || !G || !M || !V || rows <= 0 || d <= 0 || step < 1) return MSAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool renorm = renorm_eps >= 0.f;
  if ((d & 3) != 0 || d > 8192) {                  // shapes the fused kernel does not take: the separate passes
    int rc = msae_adam_rows_f32(W, G, M, V, rows, d, total_sumsq, max_norm, project, lr, beta1, beta2, eps, step, stream);
    if (rc) return rc;
    if (renorm) { rc = msae_unit_norm_rows_f32(W, rows, d, renorm_eps, stream); if (rc) return rc; }
    if (prepared) return T_next > 0 ? msae_encoder_refresh_for(W, rows, d, prepared, T_next, opts, stream)
                                    : msae_encoder_refresh(W, rows, d, prepared, opts, stream);
    return 0;
  }
  if (!(msae_aligned(W, 16) && msae_aligned(G, 16) && msae_aligned(M, 16) && msae_aligned(V, 16))) return MSAE_EALIGN;
  AdamArgs a{};
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.project = project ? 1 : 0;
  FusedTail ft{};
  ft.renorm_eps = renorm ? renorm_eps : -1.f;
  int refresh = 0;
  if (prepared) {
    CallOpts co;
    if (!resolve_opts(opts, co)) return MSAE_EINVAL;
    if (!msae_aligned(prepared, 256)) return MSAE_EALIGN;
    const int N = rows;
    Prepared p = make_prepared(N, d);
    if (p.S) {
      const bool i8 = co.mode == 1 && i8_shape_ok(N, d);
      const int modes = (i8 ? 2 : 1) | (T_next > 256 ? 4 : 0);     // as msae_encoder_refresh[_for]
      p.valid = prep_valid_bits(modes, N, d);
      p.dseed = i8 ? co.seed : 0ull;                                // (the shared dither of the int8 operands, encode_defs.h)
      MSAE_HIP_TRY(hipMemcpyAsync(prepared, &p, sizeof(p), hipMemcpyHostToDevice, s));
      unsigned char *base = static_cast<unsigned char *>(prepared);
      ft.rq = row_quant_out(base, p, modes, i8);
      ft.rq.seed = co.seed;                                        // msae_options::dither: this refresh's own seed
      if (p.dseed != 0ull)
        hipLaunchKernelGGL(sd_table_kernel, dim3(1), dim3(1024), 0, s, co.seed, d, reinterpret_cast<int *>(base + p.off_sdtab));
      ft.wb = reinterpret_cast<unsigned short *>(base + p.off_wb);
      ft.ws = reinterpret_cast<unsigned short *>(base + p.off_ws);
      refresh = i8 ? 1 : 2;
    } else {
      MSAE_HIP_TRY(hipMemcpyAsync(prepared, &p, sizeof(p), hipMemcpyHostToDevice, s));   // no fused path for this shape: header only
    }
  }
  if (refresh == 1)
    hipLaunchKernelGGL(adam_rows_fused_kernel<1>, dim3(rows), dim3(256), 0, s, W, G, M, V, d, total_sumsq, a, ft);
  else if (refresh == 2)
    hipLaunchKernelGGL(adam_rows_fused_kernel<2>, dim3(rows), dim3(256), 0, s, W, G, M, V, d, total_sumsq, a, ft);
  else
    hipLaunchKernelGGL(adam_rows_fused_kernel<0>, dim3(rows), dim3(256), 0, s, W, G, M, V, d, total_sumsq, a, ft);
  return msae_launch_status();
}
