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
template <typename IT>
__global__ __launch_bounds__(DEC_THREADS) void decode_fwd_v4_kernel(
    const IT *__restrict__ idx, const float *__restrict__ acts,
    const float *__restrict__ W_dec, const float *__restrict__ b_dec, float *__restrict__ out,
    int k, int N, int d, int32_t *status) {
  const int a = blockIdx.x;
  const int col = (blockIdx.y * DEC_THREADS + threadIdx.x) * 4;
  const IT *ip = idx + (size_t)a * k;
  const float *vp = acts + (size_t)a * k;
  if (col >= d) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int j = 0;
  for (; j + DEC_UNROLL <= k; j += DEC_UNROLL) {
    f32x4 w[DEC_UNROLL];
    float v[DEC_UNROLL];
#pragma unroll
    for (int u = 0; u < DEC_UNROLL; ++u) {
      const IT iw = ip[j + u];
      int i = (int)iw;
      v[u] = vp[j + u];
      const bool bad = iw < 0 || iw >= (IT)N;
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
    const IT iw = ip[j];
    int i = (int)iw;
    float v = vp[j];
    if (iw < 0 || iw >= (IT)N) {
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
template <typename IT>
__global__ __launch_bounds__(DEC_THREADS) void decode_fwd_scalar_kernel(
    const IT *__restrict__ idx, const float *__restrict__ acts,
    const float *__restrict__ W_dec, const float *__restrict__ b_dec, float *__restrict__ out,
    int k, int N, int d, int32_t *status) {
  const int a = blockIdx.x;
  const int col = blockIdx.y * DEC_THREADS + threadIdx.x;
  if (col >= d) return;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const IT iw = idx[(size_t)a * k + j];
    int i = (int)iw;
    float v = acts[(size_t)a * k + j];
    if (iw < 0 || iw >= (IT)N) {
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
    const float *__restrict__ W_dec, int A, int k, int N, int d, float *__restrict__ g_acts,
    int32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)A * k) return;
  const int a = (int)(pair / k);
  const int i = idx[pair];
  if ((unsigned)i >= (unsigned)N) {     // wave-uniform.  kernels.py:389 device_assert: no gradient, flagged
    if (lane == 0) {
      g_acts[pair] = 0.f;
      if (status) atomicOr(status, 1);
    }
    return;
  }
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

// ---- g_W[n][:] = sum over pairs (a, j) with idx[a][j] == n of acts[a][j] * grad_out[a][:] -------------
// Replaces triton_sparse_transpose_dense_matmul (kernels.py:10-175: torch.sort of the A*k indices,
// a 2-GiB zeros() and run-length tl.atomic_add).  Here: counting sort of the pairs by feature
// (histogram -> exclusive scan -> fill), then ONE wave per feature row accumulates its pairs in
// registers in ascending pair order and writes the row once -- no atomics on the 2-GiB gradient,
// no zero-fill (rows without pairs are written as zeros), and a fixed summation order (for rows
// with at most 64 pairs), so the weight gradients are reproducible run to run.  HBM-bound: N*d*4 B written + (A*k)*d*4 B of grad_out
// rows read (mostly L2 / Infinity-Cache hits: grad_out is A*d*4 B).
__global__ __launch_bounds__(256) void wgrad_count_kernel(const int32_t *__restrict__ idx,
                                                          const float *__restrict__ acts, long pairs,
                                                          int N, int *__restrict__ counts,
                                                          int32_t *__restrict__ status) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const int i = idx[p];
  if ((unsigned)i >= (unsigned)N) {      // kernels.py:28-36 asserts on the host; here: dropped and flagged
    if (status) atomicOr(status, 1);
    return;
  }
  if (acts[p] != 0.f) atomicAdd(counts + i, 1);
}

// single workgroup: offsets[n] = exclusive prefix of counts; cursor = copy; offsets[N] = total.  Tiles of 8192 counts: a thread
// owns 8 CONSECUTIVE counts (two coalesced 16-B loads, its local prefix in registers), one wave scan of the thread totals, one
// LDS hop across the 16 waves, coalesced 16-B stores: 16 iterations at N = 131072 (the 1024-wide scan of round 3 took 128
// iterations = 137 us; a chunk-per-thread scan, tried first in round 4, 265 us: 4-byte accesses 512 B apart).
__global__ __launch_bounds__(1024) void wgrad_scan_kernel(const int *__restrict__ counts, int N,
                                                          int *__restrict__ offsets,
                                                          int *__restrict__ cursor) {
  __shared__ int wave_tot[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 8192) {
    const int i0 = base + threadIdx.x * 8;
    int v[8];
    if (i0 + 8 <= N) {
      const i32x4 a = *reinterpret_cast<const i32x4 *>(counts + i0), b = *reinterpret_cast<const i32x4 *>(counts + i0 + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = i0 + e < N ? counts[i0 + e] : 0;
    }
    int tot = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int t = v[e]; v[e] = tot; tot += t; }   // v[e] = exclusive prefix inside the thread
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < wave; ++w) pre += wave_tot[w];
    const int excl = pre + incl - tot;
    if (i0 + 8 <= N) {
      const i32x4 a = {excl + v[0], excl + v[1], excl + v[2], excl + v[3]}, b = {excl + v[4], excl + v[5], excl + v[6], excl + v[7]};
      *reinterpret_cast<i32x4 *>(offsets + i0) = a; *reinterpret_cast<i32x4 *>(offsets + i0 + 4) = b;
      *reinterpret_cast<i32x4 *>(cursor + i0) = a; *reinterpret_cast<i32x4 *>(cursor + i0 + 4) = b;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (i0 + e < N) { offsets[i0 + e] = excl + v[e]; cursor[i0 + e] = excl + v[e]; }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = pre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[N] = carry;
}

__global__ __launch_bounds__(256) void wgrad_fill_kernel(const int32_t *__restrict__ idx,
                                                         const float *__restrict__ acts, long pairs,
                                                         int N, int *__restrict__ cursor,
                                                         int *__restrict__ perm) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const int i = idx[p];
  if ((unsigned)i < (unsigned)N && acts[p] != 0.f) perm[atomicAdd(cursor + i, 1)] = (int)p;
}

// Ascending sort of a[0..L) by ONE wave with a network whose comparators all point the same way (first
// step of every merge stage mirrors the block, the rest are half-cleaners), so the slots >= L act as +inf
// without being stored.  `sync()` orders one step's writes before the next step's reads.
template <class Sync>
__device__ __forceinline__ void wave_sort_asc(int *a, int L, int lane, Sync &&sync) {
  const int np = next_pow2(L);
  for (int size = 2; size <= np; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const bool flip = stride == (size >> 1);
      for (int i = lane; i < (np >> 1); i += 64) {
        const int blk = i / stride, off = i % stride;
        const int lo = blk * (stride << 1) + off;
        const int hi = flip ? blk * size + size - 1 - off : lo + stride;
        if (hi < L) {
          const int x = a[lo], y = a[hi];
          if (x > y) { a[lo] = y; a[hi] = x; }
        }
      }
      sync();
    }
  }
}

constexpr int WGRAD_LDS_SEG = 1024;   // pair ids a wave can sort in its LDS slice

// one wave per feature row n; lane owns float4 columns lane, lane+64, ... ; d % 4 == 0
__global__ __launch_bounds__(256) void wgrad_accum_kernel(const float *__restrict__ acts,
                                                          const float *__restrict__ grad_out,
                                                          const int *__restrict__ offsets,
                                                          int *__restrict__ perm, int k, int N, int d,
                                                          float *__restrict__ g_W, float *__restrict__ rowsq,
                                                          float *__restrict__ rowsum) {
  __shared__ int s_seg[4][WGRAD_LDS_SEG];
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int beg = offsets[n], end = offsets[n + 1];
  const int L = end - beg;
  // deterministic summation order: the segment (filled through an atomic cursor, i.e. in any order) is sorted
  // by pair id = by token.  <= 64 pairs (nearly every feature): one pair id per lane, a 21-step bitonic network over the wave's
  // registers (round 4; lane 0's insertion sort in global memory was a chain of dependent L2 round trips per pair -- most of the
  // kernel's time); <= 1024: the wave, in LDS; longer (a feature active on more than 1024 tokens of the call): the wave, in
  // place, agent-scope fences between the steps.
  int my_p = 0x7FFFFFFF;                               // lane l's pair id of the (sorted) first 64
  if (L <= 64) {
    if (lane < L) my_p = perm[beg + lane];
    if (L > 1) {
#pragma unroll
      for (int kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
          const int other = __shfl_xor(my_p, j, 64);
          const bool keep_min = ((lane & kk) == 0) == ((lane & j) == 0);
          my_p = keep_min ? (my_p < other ? my_p : other) : (my_p > other ? my_p : other);
        }
    }
  } else if (L > 64 && L <= WGRAD_LDS_SEG) {
    int *seg = s_seg[threadIdx.x >> 6];
    for (int i = lane; i < L; i += 64) seg[i] = perm[beg + i];
    __builtin_amdgcn_wave_barrier();
    wave_sort_asc(seg, L, lane, [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); });
    for (int i = lane; i < L; i += 64) perm[beg + i] = seg[i];
  } else if (L > WGRAD_LDS_SEG) {
    volatile int *seg = perm + beg;
    wave_sort_asc(const_cast<int *>(seg), L, lane, [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); __builtin_amdgcn_wave_barrier(); });
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float ss = 0.f;                                      // this lane's share of |g_W[n]|^2 (rowsq)
  // the segment's first 64 (pair id, activation) in registers, one per lane: the pair loop below reads them with readlane
  // (scalar operands: the row address of a pair no longer hangs on two dependent vector loads per pass, and the loads of
  // several pairs are in flight together)
  float my_v = 0.f;
  if (L > 64) my_p = perm[beg + lane];                 // (sorted in memory above)
  if (lane < L) my_v = acts[my_p];
  for (int c0 = lane * 4; c0 < d; c0 += 256 * 4) {     // 4 column chunks in flight per pass
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Lr = L < 64 ? L : 64;
    int e = 0;
    for (; e + 2 <= Lr; e += 2) {                       // two pairs' rows in flight; the fma chain stays in ascending pair order
      const int p0 = __builtin_amdgcn_readlane(my_p, e), p1 = __builtin_amdgcn_readlane(my_p, e + 1);
      const float v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), e));
      const float v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), e + 1));
      const float *g0 = grad_out + (size_t)(p0 / k) * d, *g1 = grad_out + (size_t)(p1 / k) * d;
      f32x4 a0[4], a1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256;
        a0[u] = c < d ? *reinterpret_cast<const f32x4 *>(g0 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        a1[u] = c < d ? *reinterpret_cast<const f32x4 *>(g1 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u][0] = __builtin_fmaf(v0, a0[u][0], acc[u][0]); acc[u][1] = __builtin_fmaf(v0, a0[u][1], acc[u][1]);
        acc[u][2] = __builtin_fmaf(v0, a0[u][2], acc[u][2]); acc[u][3] = __builtin_fmaf(v0, a0[u][3], acc[u][3]);
        acc[u][0] = __builtin_fmaf(v1, a1[u][0], acc[u][0]); acc[u][1] = __builtin_fmaf(v1, a1[u][1], acc[u][1]);
        acc[u][2] = __builtin_fmaf(v1, a1[u][2], acc[u][2]); acc[u][3] = __builtin_fmaf(v1, a1[u][3], acc[u][3]);
      }
    }
    for (; e < L; ++e) {                                // the odd pair, and pairs beyond the 64 held in registers
      int pe; float ve;
      if (e < 64) { pe = __builtin_amdgcn_readlane(my_p, e); ve = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), e)); }
      else { pe = perm[beg + e]; ve = acts[pe]; }
      const float *g = grad_out + (size_t)(pe / k) * d;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256;
        if (c < d) {
          const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + c);
          acc[u][0] = __builtin_fmaf(ve, gv[0], acc[u][0]); acc[u][1] = __builtin_fmaf(ve, gv[1], acc[u][1]);
          acc[u][2] = __builtin_fmaf(ve, gv[2], acc[u][2]); acc[u][3] = __builtin_fmaf(ve, gv[3], acc[u][3]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * 256;
      if (c < d) {
        *reinterpret_cast<f32x4 *>(g_W + (size_t)n * d + c) = acc[u];
        ss += acc[u][0] * acc[u][0] + acc[u][1] * acc[u][1] + acc[u][2] * acc[u][2] + acc[u][3] * acc[u][3];
      }
    }
  }
  if (rowsum) {  // sum of the row's pair activations in ascending pair order (lane l: pairs l, l + 64, ...; then a fixed
                 // reduction tree): with acts = the latents' gradients this IS the encoder bias gradient of feature n -- no
                 // index_add_ with its atomics, bit-reproducible
    float sa = my_v;                                     // (pair lane of the sorted first 64; 0 beyond the segment)
    for (int e = beg + 64 + lane; e < end; e += 64) sa += acts[perm[e]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sa += __shfl_xor(sa, off, 64);
    if (lane == 0) rowsum[n] = sa;
  }
  if (rowsq) {   // the gradient-norm pass of clip_grad_norm_ (trainer.py:390) for free: the row is in registers, fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) rowsq[n] = ss;
  }
}

__global__ void wgrad_zero_kernel(int *p, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0;
}

}  // namespace

template <typename IT>
static int decode_launch(const IT *idx, const float *acts, const float *W_dec, const float *b_dec, int A, int k,
                         int N, int d, float *out, int32_t *status, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0) return MSAE_EINVAL;
  if (A == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (d % 4 == 0) && msae_aligned(W_dec, 16) && msae_aligned(out, 16) &&
                   (!b_dec || msae_aligned(b_dec, 16));
  if (vec) {
    dim3 grid(A, (d + DEC_THREADS * 4 - 1) / (DEC_THREADS * 4));
    hipLaunchKernelGGL(decode_fwd_v4_kernel<IT>, grid, dim3(DEC_THREADS), 0, s, idx, acts, W_dec, b_dec,
                       out, k, N, d, status);
  } else {
    dim3 grid(A, (d + DEC_THREADS - 1) / DEC_THREADS);
    hipLaunchKernelGGL(decode_fwd_scalar_kernel<IT>, grid, dim3(DEC_THREADS), 0, s, idx, acts, W_dec,
                       b_dec, out, k, N, d, status);
  }
  return msae_launch_status();
}

extern "C" int msae_decode_f32(const int32_t *idx, const float *acts, const float *W_dec,
                               const float *b_dec, int A, int k, int N, int d, float *out,
                               int32_t *status, void *stream) {
  return decode_launch<int32_t>(idx, acts, W_dec, b_dec, A, k, N, d, out, status, stream);
}

// same with the index width of Tensor.topk (int64): what the reference hands to decoder_impl
extern "C" int msae_decode_i64_f32(const int64_t *idx, const float *acts, const float *W_dec,
                                   const float *b_dec, int A, int k, int N, int d, float *out,
                                   int32_t *status, void *stream) {
  return decode_launch<int64_t>(idx, acts, W_dec, b_dec, A, k, N, d, out, status, stream);
}

extern "C" int msae_decode_bwd_acts_f32(const int32_t *idx, const float *grad_out,
                                        const float *W_dec, int A, int k, int N, int d,
                                        float *g_acts, int32_t *status, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0) return MSAE_EINVAL;
  if (A == 0) return 0;
  if (d % 4 == 0 && (!msae_aligned(W_dec, 16) || !msae_aligned(grad_out, 16))) return MSAE_EALIGN;
  const long pairs = (long)A * k;
  hipLaunchKernelGGL(decode_bwd_acts_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, idx, grad_out, W_dec, A, k, N, d, g_acts, status);
  return msae_launch_status();
}

extern "C" size_t msae_decode_bwd_wdec_ws_bytes(int A, int k, int N) {
  if (A < 0 || k <= 0 || N <= 0) return 0;
  return msae_align_up((size_t)(N + 1) * 4, 256) * 3 + msae_align_up((size_t)A * k * 4, 256);
}

extern "C" int msae_decode_bwd_wdec_f32(const int32_t *idx, const float *acts,
                                        const float *grad_out, int A, int k, int N, int d,
                                        float *g_W_dec, float *row_sumsq, float *row_act_sum, int32_t *status, void *ws,
                                        size_t ws_bytes, void *stream) {
  if (A < 0 || k <= 0 || N <= 0 || d <= 0 || d % 4 != 0) return MSAE_EINVAL;
  if (!ws || ws_bytes < msae_decode_bwd_wdec_ws_bytes(A, k, N)) return MSAE_EWS;
  if (!msae_aligned(grad_out, 16) || !msae_aligned(g_W_dec, 16) || !msae_aligned(ws, 256)) return MSAE_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const size_t seg = msae_align_up((size_t)(N + 1) * 4, 256);
  int *counts = reinterpret_cast<int *>(ws);
  int *offsets = reinterpret_cast<int *>(static_cast<unsigned char *>(ws) + seg);
  int *cursor = reinterpret_cast<int *>(static_cast<unsigned char *>(ws) + 2 * seg);
  int *perm = reinterpret_cast<int *>(static_cast<unsigned char *>(ws) + 3 * seg);
  const long pairs = (long)A * k;
  hipLaunchKernelGGL(wgrad_zero_kernel, dim3(256), dim3(256), 0, s, counts, N + 1);
  if (pairs > 0) {
    const unsigned pb = (unsigned)((pairs + 255) / 256);
    hipLaunchKernelGGL(wgrad_count_kernel, dim3(pb), dim3(256), 0, s, idx, acts, pairs, N, counts, status);
    hipLaunchKernelGGL(wgrad_scan_kernel, dim3(1), dim3(1024), 0, s, counts, N, offsets, cursor);
    hipLaunchKernelGGL(wgrad_fill_kernel, dim3(pb), dim3(256), 0, s, idx, acts, pairs, N, cursor, perm);
  } else {
    hipLaunchKernelGGL(wgrad_zero_kernel, dim3(256), dim3(256), 0, s, offsets, N + 1);
  }
  hipLaunchKernelGGL(wgrad_accum_kernel, dim3((N + 3) / 4), dim3(256), 0, s, acts, grad_out, offsets, perm,
                     k, N, d, g_W_dec, row_sumsq, row_act_sum);
  return msae_launch_status();
}
