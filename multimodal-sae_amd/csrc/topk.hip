// topk.hip -- canonical top-k of dense f32 rows (value descending, ties by ascending index).
//
// Replaces Tensor.topk as used by Sae.select_topk (reference sae/sae.py:179-181) and by the cache
// loop (features/cache.py:210-212).  The fused encoder (encode_fused.hip) never materialises the
// dense [T][N] latents; this kernel serves the dense legacy API, the exact fallback, and the
// sample-threshold selection of the fused encoder.
//
// One 1024-thread workgroup per row.  Radix select on the order-preserving u32 key of each value
// (12 + 12 + 8 bits, LDS histograms) finds the key of the k-th largest element and how many
// elements equal to it are needed (r); an index-ordered collect pass (block prefix sums) gathers
// the k winners -- every element above the pivot plus the r lowest-index elements equal to it --
// and an LDS bitonic sort puts them in canonical order.  HBM-bound: the row is streamed once from
// HBM and re-read from L2 by the later passes (N*4 B = 512 KiB at N = 131072).
#include "common.h"

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int TK_BINS = 4096;

struct TkShared {
  unsigned hist[TK_BINS];
  unsigned wave_tot[TK_WAVES];
  unsigned bin;
  unsigned remaining;
  unsigned base_gt;
  unsigned base_eq;
};

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned n = __shfl_up(v, off, 64);
    if (lane >= off) v += n;
  }
  return v;
}

// Block-wide exclusive prefix sum of `v` in thread order; returns the exclusive prefix and
// writes the block total to *total.  Uses sh.wave_tot; contains two barriers.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, TkShared &sh, unsigned *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = wave_incl_scan(v, lane);
  __syncthreads();  // previous users of wave_tot are done
  if (lane == 63) sh.wave_tot[wave] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TK_WAVES; ++w) {
    unsigned t = sh.wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

template <bool VEC>
__device__ __forceinline__ void load4(const float *row, int i, int N, float (&v)[4]) {
  if constexpr (VEC) {
    f32x4 t = *reinterpret_cast<const f32x4 *>(row + i);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i + e < N) ? row[i + e] : 0.f;
  }
}

// One radix pass: histogram of ((key >> shift) & (nbins-1)) over elements whose key matches
// `prefix` under `mask`; then locate the bin holding the `remaining`-th largest such element.
template <bool VEC>
__device__ __forceinline__ void radix_pass(const float *row, int N, unsigned prefix, unsigned mask,
                                           int shift, int nbins, TkShared &sh) {
  const unsigned need = sh.remaining;  // read before any barrier: the winner rewrites it below
  for (int b = threadIdx.x; b < nbins; b += TK_THREADS) sh.hist[b] = 0;
  __syncthreads();
  const int n4 = (N + 3) & ~3;
  for (int i = threadIdx.x * 4; i < n4; i += TK_THREADS * 4) {
    float v[4];
    load4<VEC>(row, i, N, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (VEC || i + e < N) {
        unsigned key = f32_order_key(v[e]);
        if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & (unsigned)(nbins - 1)], 1u);
      }
    }
  }
  __syncthreads();
  // suffix search from the top bin: thread t owns bins [t*per, t*per+per)
  const int per = (nbins + TK_THREADS - 1) / TK_THREADS;  // 4 for 4096 bins, 1 for <= 1024
  // reversed thread order so an exclusive PREFIX scan over rt is a SUFFIX sum over bins
  const int rt = TK_THREADS - 1 - threadIdx.x;  // rt == 0 owns the highest bins
  unsigned local = 0;
  for (int e = 0; e < per; ++e) {
    int b = nbins - 1 - (rt * per + e);
    if (b >= 0) local += sh.hist[b];
  }
  // scan in rt order: emulate by scanning in thread order of a mirrored value
  // (thread x holds value for rt = 1023 - x; we need prefix over rt, i.e. suffix over x)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned n = __shfl_down(incl, off, 64);
    if (lane + off < 64) incl += n;
  }
  __syncthreads();
  if (lane == 0) sh.wave_tot[wave] = incl;  // total of this wave
  __syncthreads();
  unsigned above = 0;
  for (int w = wave + 1; w < TK_WAVES; ++w) above += sh.wave_tot[w];
  above += incl - local;  // elements in bins strictly above this thread's bins
  if (above < need && need <= above + local) {
    unsigned cum = above;
    for (int e = 0; e < per; ++e) {
      int b = nbins - 1 - (rt * per + e);
      if (b < 0) break;
      unsigned c = sh.hist[b];
      if (cum + c >= need) {
        sh.bin = (unsigned)b;
        sh.remaining = need - cum;  // safe: exactly one thread satisfies the outer condition
        break;
      }
      cum += c;
    }
  }
  __syncthreads();
}

template <bool VEC>
__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(const float *__restrict__ latents,
                                                               int T, int N, int k, int ld,
                                                               const int *__restrict__ n_rows,
                                                               float *__restrict__ vals,
                                                               int32_t *__restrict__ idx, TopkExtra ex) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (n_rows) T = min(T, *n_rows);         // device-side row count (exact fallback)
  TkShared &sh = *reinterpret_cast<TkShared *>(smem_raw);
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem_raw + sizeof(TkShared));
  const int kp = next_pow2(k);
  for (int rowi = blockIdx.x; rowi < T; rowi += gridDim.x) {   // rows blockIdx.x, +gridDim.x, ...
  const float *row = latents + (size_t)rowi * ld;

  if (threadIdx.x == 0) {
    sh.remaining = (unsigned)k;
    sh.base_gt = 0;
    sh.base_eq = 0;
  }
  for (int i = threadIdx.x; i < kp; i += TK_THREADS) keys[i] = 0ull;
  __syncthreads();

  // ---- radix select of the pivot key -------------------------------------------------------
  unsigned prefix = 0, mask = 0;
  radix_pass<VEC>(row, N, prefix, mask, 20, 4096, sh);
  prefix |= sh.bin << 20; mask |= 0xFFFu << 20;
  radix_pass<VEC>(row, N, prefix, mask, 8, 4096, sh);
  prefix |= sh.bin << 8; mask |= 0xFFFu << 8;
  radix_pass<VEC>(row, N, prefix, mask, 0, 256, sh);
  prefix |= sh.bin;
  const unsigned pivot = prefix;           // key of the k-th largest element
  const unsigned r = sh.remaining;         // how many elements == pivot to take (lowest index)
  const unsigned n_gt = (unsigned)k - r;   // elements strictly above the pivot
  __syncthreads();

  // ---- index-ordered collect ---------------------------------------------------------------
  const int n4 = (N + 3) & ~3;
  for (int base = 0; base < n4; base += TK_THREADS * 4) {
    const int i = base + threadIdx.x * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned key[4];
    unsigned c_gt = 0, c_eq = 0;
    if (i < n4) load4<VEC>(row, i, N, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = (i + e < N);
      key[e] = in ? f32_order_key(v[e]) : 0u;
      c_gt += (in && key[e] > pivot);
      c_eq += (in && key[e] == pivot);
    }
    const unsigned packed = c_gt | (c_eq << 16);
    if (__syncthreads_or((int)packed) == 0) continue;
    unsigned total;
    const unsigned ex = block_excl_scan(packed, sh, &total);
    unsigned p_gt = sh.base_gt + (ex & 0xFFFFu);
    unsigned p_eq = sh.base_eq + (ex >> 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i + e < N) {
        if (key[e] > pivot) {
          keys[p_gt++] = rank_key(v[e], i + e);
        } else if (key[e] == pivot) {
          if (p_eq < r) keys[n_gt + p_eq] = rank_key(v[e], i + e);
          ++p_eq;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      sh.base_gt += total & 0xFFFFu;
      sh.base_eq += total >> 16;
    }
    __syncthreads();
    if (sh.base_gt >= n_gt && sh.base_eq >= r) break;  // uniform: all k winners found
  }
  __syncthreads();

  // ---- canonical order ---------------------------------------------------------------------
  bitonic_sort_desc_u64(keys, kp);
  const int orow = ex.row_map ? ex.row_map[rowi] : rowi;
  for (int j = threadIdx.x; j < k; j += TK_THREADS) {
    const unsigned long long kk = keys[j];
    const int ix = rank_key_index(kk);
    if (idx) idx[(size_t)orow * k + j] = ix;
    if (ex.idx64) ex.idx64[(size_t)orow * k + j] = ix;
    vals[(size_t)orow * k + j] = row[ix];  // the stored value (keeps -0.0 as stored)
  }
  // a token the fused encoder could not verify, recomputed here: status 1 (msae_options::status_detail keeps why)
  if (ex.status && threadIdx.x == 0) ex.status[orow] = ex.detail ? (1 | ((ex.status[orow] & ~3) << 8)) : 1;
  __syncthreads();   // sh / keys are reused by the next row
  }
}

// ---- r-th largest value of each row (threshold select of the fused encoder) ---------------------
// One wave per row; the row lives in registers (VPL values per lane) and the r-th largest order
// key is found by bisection on the key space (count(key >= mid) by lane + wave reduce), VALU-bound.
// The caller needs a THRESHOLD with at least r values at or above it, not the exact order statistic
// (the fused encoder verifies its candidates afterwards), so the bisection stops after the sign,
// the exponent and 9 mantissa bits: the result is the r-th largest rounded DOWN by < 0.2 %
// (18 steps instead of 32).
constexpr int KTH_LOW_BIT = 14;
template <int VPL>
__global__ __launch_bounds__(256) void kth_value_kernel(const float *__restrict__ rows, int T, int S,
                                                        int ld, int r, float *__restrict__ out,
                                                        int out_ld, int out_col, KthPush push) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float *row = rows + (size_t)t * ld;
  unsigned key[VPL];
#pragma unroll
  for (int i = 0; i < VPL / 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (i * 64 + lane) * 4);
    key[4 * i + 0] = f32_order_key(v[0]); key[4 * i + 1] = f32_order_key(v[1]);
    key[4 * i + 2] = f32_order_key(v[2]); key[4 * i + 3] = f32_order_key(v[3]);
  }
  unsigned lo = 0u;               // invariant: count(key >= lo) >= r
  int first_bit = 31;
  {  // The values of a row share their sign and most of their exponent: if r of them reach the row maximum's top 8 bits, those 8
     // bits ARE the answer's (no larger prefix has any value at all) and the bisection starts below them -- 12 steps instead of 18.
    unsigned mx = 0u;
#pragma unroll
    for (int i = 0; i < VPL; ++i) mx = key[i] > mx ? key[i] : mx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
    const unsigned pre = mx & 0xFF000000u;
    int c = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) c += (key[i] >= pre) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (c >= r) { lo = pre; first_bit = 23; }
  }
#pragma unroll 1
  for (int bit = first_bit; bit >= KTH_LOW_BIT; --bit) {
    const unsigned mid = lo | (1u << bit);
    int c = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) c += (key[i] >= mid) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (c >= r) lo = mid;
  }
  if (lane == 0) out[(size_t)t * out_ld + out_col] = f32_from_order_key(lo);
  if (push.cnt && f32_from_order_key(lo) > 0.f) {      // the row's values above the threshold -> its candidate list
    // one list reservation per row (a returning atomic per hit would chain ~r round trips per wave);
    // lane masks of the compares on the scalar unit (the keys fill the registers: no per-lane counters or addresses)
    int total = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] > lo));
      if ((i & 7) == 7) asm volatile("" : "+s"(total));   // at most 8 lane masks in flight (64 pairs would spill SGPRs into VGPRs)
    }
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(push.cnt + (size_t)t * push.cnt_stride, total);
    base = __builtin_amdgcn_readfirstlane(base);
    unsigned long long *list = push.cand + (size_t)t * (push.row_stride ? push.row_stride : push.cap);
    unsigned lo2 = lo;
    asm volatile("" : "+v"(lo2));                      // the masks are recomputed here, not kept (64 SGPR pairs would spill into VGPRs)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const bool hit = key[i] > lo2;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
      if (m == 0ull) continue;                         // wave-uniform
      if (hit) {
        const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const int j = ((i >> 2) * 64 + lane) * 4 + (i & 3);
        // the key goes through a real move: otherwise every key lives in the high half of a register PAIR reserved for this
        // 64-bit store for the whole kernel (2 x VPL VGPRs: half the occupancy)
        unsigned kk;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kk) : "v"(key[i]));
        if (slot < push.cap) list[slot] = ((unsigned long long)kk << 32) | (unsigned)(0x7FFFFFFF - (j * push.stride + push.off));
      }
      base += __builtin_popcountll(m);
      if ((i & 7) == 7) asm volatile("" : "+s"(base));
    }
  }
}

// ---- merge of per-shard top-k_loc lists (feature-sharded encode, msae/parallel.py) ---------------
// gathered: int32 [G][2][T][kl] exactly as all_gather_into_tensor lays out each rank's packed
// [2][T][kl] block (plane 0 = f32 activation bits, plane 1 = GLOBAL feature index).
// One wave per token: G*kl rank keys into LDS, bitonic sort, canonical top-k out.  flagged[t] = 1
// when some shard's LAST gathered latent ranks inside the merged top-k (that shard may own more
// members than it sent; the host redoes those tokens with k_loc = k).
__global__ __launch_bounds__(64) void merge_topk_kernel(const int32_t *__restrict__ gathered, int T,
                                                        int G, int kl, int k,
                                                        float *__restrict__ vals,
                                                        int32_t *__restrict__ idx,
                                                        int32_t *__restrict__ flagged) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long mkeys[];
  const int lane = threadIdx.x, t = blockIdx.x;
  const int M = G * kl, np = next_pow2(M);
  unsigned long long worst_last = 0ull;  // best (largest) key among the shards' last entries
  for (int i = lane; i < np; i += 64) {
    unsigned long long key = 0ull;
    if (i < M) {
      const int g = i / kl, j = i % kl;
      const float v = __int_as_float(gathered[(((size_t)g * 2 + 0) * T + t) * kl + j]);
      const int f = gathered[(((size_t)g * 2 + 1) * T + t) * kl + j];
      key = rank_key(v, f);
      if (j == kl - 1) worst_last = key > worst_last ? key : worst_last;
    }
    mkeys[i] = key;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(worst_last, off, 64);
    worst_last = o > worst_last ? o : worst_last;
  }
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = lane; i < (np >> 1); i += 64) {
        const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = mkeys[lo], y = mkeys[hi];
        if ((x < y) == desc) { mkeys[lo] = y; mkeys[hi] = x; }
      }
    }
  __syncthreads();
  for (int j = lane; j < k; j += 64) {
    const unsigned long long key = mkeys[j];
    idx[(size_t)t * k + j] = rank_key_index(key);
    vals[(size_t)t * k + j] = f32_from_order_key((unsigned)(key >> 32));
  }
  if (lane == 0 && flagged) flagged[t] = (kl < k && worst_last >= mkeys[k - 1]) ? 1 : 0;
}

// rows[0 .. n) = the t with flags[t] != 0, ascending; *n_rows = n.  ONE workgroup walks the flags 1024 at a time (ballot
// prefix counts): the device-side redo list of the feature-sharded engine's second round -- every rank derives the same
// list from the same gathered data, nothing is read back to the host.
__global__ __launch_bounds__(1024) void compact_flags_kernel(const int32_t *__restrict__ flags, int T,
                                                             int32_t *__restrict__ rows, int32_t *__restrict__ n_rows) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int t = base + tid;
    const bool f = t < T && flags[t] != 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
    if (lane == 0) wsum[wv] = __builtin_popcountll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (f) rows[off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = t;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) *n_rows = s_base;
}

// merge_topk_kernel for the tokens of a mask only (second round: the redone tokens' full local top-k lists replace the
// truncated merge; everybody else keeps round 1's result).  One wave per token, same key order.
__global__ __launch_bounds__(64) void merge_topk_masked_kernel(const int32_t *__restrict__ gathered, int T, int G, int kl,
                                                               int k, const int32_t *__restrict__ mask,
                                                               float *__restrict__ vals, int32_t *__restrict__ idx,
                                                               int64_t *__restrict__ idx64) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long mkeys[];
  const int lane = threadIdx.x, t = blockIdx.x;
  if (mask[t] == 0) return;                         // wave-uniform
  const int M = G * kl, np = next_pow2(M);
  for (int i = lane; i < np; i += 64) {
    unsigned long long key = 0ull;
    if (i < M) {
      const int g = i / kl, j = i % kl;
      const float v = __int_as_float(gathered[(((size_t)g * 2 + 0) * T + t) * kl + j]);
      key = rank_key(v, gathered[(((size_t)g * 2 + 1) * T + t) * kl + j]);
    }
    mkeys[i] = key;
  }
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = lane; i < (np >> 1); i += 64) {
        const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = mkeys[lo], y = mkeys[hi];
        if ((x < y) == desc) { mkeys[lo] = y; mkeys[hi] = x; }
      }
    }
  __syncthreads();
  for (int j = lane; j < k; j += 64) {
    const unsigned long long key = mkeys[j];
    if (idx) idx[(size_t)t * k + j] = rank_key_index(key);
    if (idx64) idx64[(size_t)t * k + j] = rank_key_index(key);
    vals[(size_t)t * k + j] = f32_from_order_key((unsigned)(key >> 32));
  }
}

}  // namespace

extern "C" size_t msae_topk_ws_bytes(int T, int N, int k) {
  (void)T; (void)N; (void)k;
  return 0;
}

// ld = row pitch in elements (>= N); exposed to the other translation units of the library.
int msae_topk_launch(const float *latents, int T, int N, int k, int ld, const int *n_rows,
                     float *vals, int32_t *idx, hipStream_t s, const TopkExtra &ex) {
  // k winners live in LDS as 64-bit rank keys: 16384 x 8 B + the histograms fit the CU's 160 KB (AuxK asks for
  // d_in / 2 latents, sae.py:209: residual streams up to 32768 wide)
  if (T < 0 || N <= 0 || k <= 0 || k > N || k > 16384 || ld < N) return MSAE_EINVAL;
  if (T == 0) return 0;
  const size_t smem = sizeof(TkShared) + (size_t)next_pow2(k) * sizeof(unsigned long long);
  const bool vec = (N % 4 == 0) && (ld % 4 == 0) && msae_aligned(latents, 16);
  if (smem > 48 * 1024) {
    if (vec) MSAE_HIP_TRY(hipFuncSetAttribute((const void *)topk_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    else MSAE_HIP_TRY(hipFuncSetAttribute((const void *)topk_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  const int grid = (n_rows && T > 128) ? 128 : T;   // device-side count: workgroups loop over the rows
  if (vec)
    hipLaunchKernelGGL(topk_rows_kernel<true>, dim3(grid), dim3(TK_THREADS), smem, s, latents, T, N, k, ld,
                       n_rows, vals, idx, ex);
  else
    hipLaunchKernelGGL(topk_rows_kernel<false>, dim3(grid), dim3(TK_THREADS), smem, s, latents, T, N, k,
                       ld, n_rows, vals, idx, ex);
  return msae_launch_status();
}

// out[t*out_ld + out_col] = a threshold with >= r values of rows[t][0..S) at or above it (the r-th
// largest rounded down by < 0.2 %), and optionally the values above it into the rows' candidate lists (KthPush); false when
// the shape has no fast kernel.
bool msae_kth_value_launch(const float *rows, int T, int S, int ld, int r, float *out, int out_ld,
                           int out_col, hipStream_t s, const KthPush &push) {
  if (S % 256 || ld % 4 || !msae_aligned(rows, 16) || r < 1 || r > S) return false;
  const dim3 grid((T + 3) / 4), block(256);
  switch (S / 64) {
#define KTH_CASE(V) case V: hipLaunchKernelGGL(kth_value_kernel<V>, grid, block, 0, s, rows, T, S, ld, r, out, out_ld, out_col, push); return true;
    KTH_CASE(4) KTH_CASE(8) KTH_CASE(16) KTH_CASE(32) KTH_CASE(64) KTH_CASE(128)
#undef KTH_CASE
    default: return false;
  }
}

extern "C" int msae_topk_f32(const float *latents, int T, int N, int k, float *vals, int32_t *idx,
                             void *ws, size_t ws_bytes, void *stream) {
  (void)ws; (void)ws_bytes;
  return msae_topk_launch(latents, T, N, k, N, nullptr, vals, idx, (hipStream_t)stream, TopkExtra());
}

extern "C" int msae_merge_topk(const int32_t *gathered, int T, int G, int kl, int k, float *vals,
                               int32_t *idx, int32_t *flagged, void *stream) {
  if (T < 0 || G <= 0 || kl <= 0 || k <= 0 || (long)G * kl < k || (long)G * kl > 8192) return MSAE_EINVAL;
  if (T == 0) return 0;
  const size_t smem = (size_t)next_pow2(G * kl) * sizeof(unsigned long long);
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)merge_topk_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(merge_topk_kernel, dim3(T), dim3(64), smem, (hipStream_t)stream, gathered, T, G,
                     kl, k, vals, idx, flagged);
  return msae_launch_status();
}

extern "C" int msae_compact_flags(const int32_t *flags, int T, int32_t *rows, int32_t *n_rows, void *stream) {
  if (T < 0 || !rows || !n_rows || (T > 0 && !flags)) return MSAE_EINVAL;
  hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flags, T, rows, n_rows);
  return msae_launch_status();
}

extern "C" int msae_merge_topk_masked(const int32_t *gathered, int T, int G, int kl, int k, const int32_t *mask,
                                      float *vals, int32_t *idx, int64_t *idx64, void *stream) {
  if (T < 0 || G <= 0 || kl <= 0 || k <= 0 || (long)G * kl < k || (long)G * kl > 8192 || !mask || !vals || (!idx && !idx64))
    return MSAE_EINVAL;
  if (T == 0) return 0;
  const size_t smem = (size_t)next_pow2(G * kl) * sizeof(unsigned long long);
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)merge_topk_masked_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(merge_topk_masked_kernel, dim3(T), dim3(64), smem, (hipStream_t)stream, gathered, T, G, kl, k, mask,
                     vals, idx, idx64);
  return msae_launch_status();
}
