// common.h -- shared device/host helpers for the gfx950 SAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msae.h"

#define MSAE_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

#define MSAE_HIP_TRY(expr)                  \
  do {                                      \
    hipError_t e__ = (expr);                \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

static inline int msae_launch_status() { return (int)hipGetLastError(); }

static inline bool msae_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

__host__ __device__ static inline size_t msae_align_up(size_t v, size_t a) {
  return (v + a - 1) / a * a;
}

// ---- element conversion ---------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even f32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) {
  return (float)__builtin_bit_cast(_Float16, b);
}

// Load 4 consecutive activations x[i..i+3] of element type DT as f32 (exact up-cast, sae.py:174).
template <int DT>
__device__ __forceinline__ f32x4 load_x4(const void *x, size_t i) {
  if constexpr (DT == MSAE_F32) {
    return *reinterpret_cast<const f32x4 *>(static_cast<const float *>(x) + i);
  } else {
    u16x4 r = *reinterpret_cast<const u16x4 *>(static_cast<const unsigned short *>(x) + i);
    f32x4 o;
    if constexpr (DT == MSAE_BF16) {
      o[0] = bf16_bits_to_f32(r[0]); o[1] = bf16_bits_to_f32(r[1]);
      o[2] = bf16_bits_to_f32(r[2]); o[3] = bf16_bits_to_f32(r[3]);
    } else {
      o[0] = f16_bits_to_f32(r[0]); o[1] = f16_bits_to_f32(r[1]);
      o[2] = f16_bits_to_f32(r[2]); o[3] = f16_bits_to_f32(r[3]);
    }
    return o;
  }
}
template <int DT>
__device__ __forceinline__ float load_x1(const void *x, size_t i) {
  if constexpr (DT == MSAE_F32) return static_cast<const float *>(x)[i];
  else if constexpr (DT == MSAE_BF16) return bf16_bits_to_f32(static_cast<const unsigned short *>(x)[i]);
  else return f16_bits_to_f32(static_cast<const unsigned short *>(x)[i]);
}

// ---- canonical ordering key: larger key == ranks earlier ------------------------------------
// order-preserving map of an f32 to u32 (-0 folded onto +0 so it ties with +0 like `==` does)
__device__ __forceinline__ unsigned f32_order_key(float f) {
  unsigned b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}
// 64-bit rank key of (value, index): value desc, then index asc  <=>  key desc
__device__ __forceinline__ unsigned long long rank_key(float v, int idx) {
  return ((unsigned long long)f32_order_key(v) << 32) | (unsigned)(0x7FFFFFFF - idx);
}
__device__ __forceinline__ int rank_key_index(unsigned long long k) {
  return 0x7FFFFFFF - (int)(unsigned)(k & 0xFFFFFFFFull);
}

// optional outputs of the row top-k launch (exact fallback of the fused encoder): 64-bit indices beside or
// instead of the 32-bit ones, output rows through a map (row i of the input -> token row_map[i]), and the
// status word of a token recomputed this way (1, or 1 | reason << 8 with `detail`)
struct TopkExtra {
  int64_t *idx64 = nullptr;
  const int *row_map = nullptr;
  int32_t *status = nullptr;
  int detail = 0;
};

// optional second output of the threshold select (msae_kth_value_launch): every value of the row ABOVE the threshold found is
// appended to its row's candidate list as (order key << 32 | 0x7FFFFFFF - feature), feature of column j = j*stride + off
struct KthPush {
  int *cnt = nullptr;                // [T] list lengths (atomically advanced); null: no push
  unsigned long long *cand = nullptr;
  int cap = 0, stride = 1, off = 0;
  int cnt_stride = 1;                // row t's counter is cnt[t * cnt_stride] ...
  int row_stride = 0;                // ... and its list starts at cand[t * row_stride] (0: cap) -- segmented lists: segment 0
};

// Bitonic sort of n (power of two) 64-bit keys in LDS, DESCENDING; all threads of the block call.
__device__ __forceinline__ void bitonic_sort_desc_u64(unsigned long long *s, int n) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
        int lo = (i / stride) * (stride << 1) + (i % stride);
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        unsigned long long a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

__host__ __device__ static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
