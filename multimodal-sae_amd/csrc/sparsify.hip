// sparsify.hip -- top-k (vals, idx) -> the reference feature cache's COO records.
//
// Replaces the dense round trip of the cache loop (reference features/cache.py:214-217:
// zeros_like + scatter_) and Cache.get_nonzeros / Cache.add (features/cache.py:42-92:
// nonzero(|x| > 1e-5), boolean gather, isin(filter), row offset).  The dense [B][S][N] tensor is
// never built: per token the k pairs are ordered by feature index (== row-major nonzero order),
// thresholded, filtered through a byte bitmap, and written at an offset given by an exclusive
// prefix sum of the per-token counts.  HBM-bound and tiny: k*8 B read, <= k*28 B written / token.
#include "common.h"

namespace {

__device__ __forceinline__ bool keep_entry(float v, int f, float thresh, const uint8_t *bitmap,
                                           int N) {
  if (!(fabsf(v) > thresh)) return false;
  if ((unsigned)f >= (unsigned)N) return false;
  if (bitmap && !bitmap[f]) return false;
  return true;
}

// one wave per token: counts_raw[t] = number of kept entries
__global__ __launch_bounds__(256) void sparsify_count_kernel(const float *__restrict__ vals,
                                                             const int32_t *__restrict__ idx,
                                                             long ntok, int k, float thresh,
                                                             const uint8_t *__restrict__ bitmap,
                                                             int N, int64_t *__restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntok) return;
  int c = 0;
  for (int j = lane; j < k; j += 64)
    c += keep_entry(vals[t * k + j], idx[t * k + j], thresh, bitmap, N) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if (lane == 0) counts[t] = c;
}

// single workgroup: in-place exclusive prefix sum of counts[0..n), total into counts[n]
__global__ __launch_bounds__(1024) void exclusive_scan_i64_kernel(int64_t *__restrict__ counts,
                                                                  long n) {
  __shared__ long long wave_tot[16];
  __shared__ long long carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long base = 0; base < n; base += 1024) {
    const long i = base + threadIdx.x;
    long long v = (i < n) ? counts[i] : 0;
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      long long o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    long long pre = carry;
    for (int w = 0; w < wave; ++w) pre += wave_tot[w];
    if (i < n) counts[i] = pre + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = pre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[n] = carry;
}

// one 64-lane workgroup per token: order by feature index, compact, write.
__global__ __launch_bounds__(64) void sparsify_write_kernel(
    const float *__restrict__ vals, const int32_t *__restrict__ idx, int S, int k, float thresh,
    const uint8_t *__restrict__ bitmap, int N, int64_t row_base, const int64_t *__restrict__ counts,
    int64_t *__restrict__ locations, float *__restrict__ activations) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const long t = blockIdx.x;
  const int kp = next_pow2(k);
  for (int j = threadIdx.x; j < kp; j += 64) {
    unsigned long long key = 0ull;
    if (j < k) {
      const float v = vals[t * k + j];
      const int f = idx[t * k + j];
      if (keep_entry(v, f, thresh, bitmap, N))
        key = ((unsigned long long)(unsigned)(0x7FFFFFFF - f) << 32) | __float_as_uint(v);
    }
    skeys[j] = key;  // key 0 (dropped) sorts last; a kept entry always has a non-zero high word
  }
  bitonic_sort_desc_u64(skeys, kp);  // descending key == ascending feature index
  const int64_t off = counts[t];
  const int n_keep = (int)(counts[t + 1] - off);
  const int64_t b = t / S, s = t % S;
  for (int j = threadIdx.x; j < n_keep; j += 64) {
    const unsigned long long key = skeys[j];
    const int f = 0x7FFFFFFF - (int)(unsigned)(key >> 32);
    locations[(off + j) * 3 + 0] = row_base + b;
    locations[(off + j) * 3 + 1] = s;
    locations[(off + j) * 3 + 2] = f;
    activations[off + j] = __uint_as_float((unsigned)(key & 0xFFFFFFFFull));
  }
}

}  // namespace

extern "C" int msae_sparsify_count(const float *vals, const int32_t *idx, int B, int S, int k,
                                   float thresh, const uint8_t *filter_bitmap, int N,
                                   int64_t *counts, void *stream) {
  if (B < 0 || S < 0 || k <= 0 || k > 4096 || N <= 0) return MSAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long ntok = (long)B * S;
  if (ntok > 0)
    hipLaunchKernelGGL(sparsify_count_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, s,
                       vals, idx, ntok, k, thresh, filter_bitmap, N, counts);
  hipLaunchKernelGGL(exclusive_scan_i64_kernel, dim3(1), dim3(1024), 0, s, counts, ntok);
  return msae_launch_status();
}

extern "C" int msae_sparsify_write(const float *vals, const int32_t *idx, int B, int S, int k,
                                   float thresh, const uint8_t *filter_bitmap, int N,
                                   int64_t row_base, const int64_t *counts, int64_t *locations,
                                   float *activations, void *stream) {
  if (B < 0 || S < 0 || k <= 0 || k > 4096 || N <= 0) return MSAE_EINVAL;
  const long ntok = (long)B * S;
  if (ntok == 0) return 0;
  const size_t smem = (size_t)next_pow2(k) * sizeof(unsigned long long);
  hipLaunchKernelGGL(sparsify_write_kernel, dim3((unsigned)ntok), dim3(64), smem,
                     (hipStream_t)stream, vals, idx, S, k, thresh, filter_bitmap, N, row_base,
                     counts, locations, activations);
  return msae_launch_status();
}
