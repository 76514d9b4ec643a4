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
// LDS rows: 36 floats, the k of a tile permuted (f_kpos): fragment reads are one ds_read_b128 per four k-steps, bank-conflict-free.  MFMA issue is the bound: 4 MFMAs (256 cycles/SIMD) per
// 4 ds_read_b32.
#include <type_traits>

#include "common.h"

namespace {

constexpr int F_BM = 128, F_BN = 128, F_BK = 32, F_PITCH = 36, F_THREADS = 256;
// LDS image of a tile row: the 32 k of a k-tile PERMUTED so that the 16 even k (the lanes 0-31 half of the MFMA's k pair) lie at
// floats [0, 16) and the 16 odd k at [16, 32): a lane fetches FOUR k-steps of its operand with one ds_read_b128 (pitch 36 floats =
// 144 B: the 16 lanes of a b128 service group hit all 64 banks once), a staging thread stores its four consecutive k as two 8-B
// pairs.  (Round 5; before: pitch 33, one ds_read2_b32 per two k-steps.)
__device__ __forceinline__ int f_kpos(int k) { return ((k & 1) << 4) + (k >> 1); }
constexpr int F_LDS_FLOATS = 2 * (F_BM + F_BN) * F_PITCH;  // double-buffered A and B tiles

// Staging registers of one k-tile: the RAW loads (x in its own element type), converted and centred only when they are
// stored to LDS behind the k-tile's MFMAs.  (Round 5: the conversion used to sit right behind each load -- `load_x4(x) - b_dec`
// -- which put an s_waitcnt vmcnt(0) behind every one of the four x loads of a k-tile, and the row gather's index load in front
// of them: ~4 exposed L2 / HBM round trips per k-tile in front of 4096 cycles of MFMA, 48 % MFMA busy by the counters
// (profiles/r05_f32_before.json).  Now the eight loads of a k-tile are issued back to back and nothing waits for them until
// the MFMAs of the current tile are done.)
template <int DT>
struct StageRegs {
  typename std::conditional<DT == MSAE_F32, f32x4, u16x4>::type a[4];   // x rows, raw
  f32x4 b[4];                                                           // W rows
  f32x4 bd;                                                             // b_dec of this thread's four k
};

// per-tile row bases of this thread's four token rows / feature rows (element offsets; rows beyond T / N are clamped to a valid
// row -- their values are zeroed at the store -- so the loads need no branches)
struct StageRows {
  size_t xo[4], wo[4];
  unsigned live_a, live_b;      // bit `it`: the row exists
};
template <int DT>
__device__ __forceinline__ StageRows stage_rows(const int *rows, int T, int d, int N, int m0, int n0) {
  const int rr = threadIdx.x >> 3;
  StageRows r;
  r.live_a = r.live_b = 0u;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    const int t = m0 + row, n = n0 + row;
    const int tc = t < T ? t : T - 1, nc = n < N ? n : N - 1;
    r.xo[it] = (rows ? (size_t)rows[tc] : (size_t)tc) * d;
    r.wo[it] = (size_t)nc * d;
    r.live_a |= (t < T ? 1u : 0u) << it;
    r.live_b |= (n < N ? 1u : 0u) << it;
  }
  return r;
}

// VEC (d % 4 == 0, aligned operands): branch-free raw loads.  k0 + q*4 may lie beyond d in the last k-tile: clamped, zeroed at
// the store.
template <int DT>
__device__ __forceinline__ void stage_load_vec(StageRegs<DT> &r, const void *x, const float *W, const float *b_dec,
                                               const StageRows &sr, int d, int k0) {
  const int q = threadIdx.x & 7;
  int kq = k0 + q * 4;
  kq = kq < d ? kq : d - 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if constexpr (DT == MSAE_F32) r.a[it] = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(x) + sr.xo[it] + kq);
    else r.a[it] = *reinterpret_cast<const u16x4 *>(static_cast<const unsigned short *>(x) + sr.xo[it] + kq);
    r.b[it] = *reinterpret_cast<const f32x4 *>(W + sr.wo[it] + kq);
  }
  r.bd = b_dec ? *reinterpret_cast<const f32x4 *>(b_dec + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int DT>
__device__ __forceinline__ void stage_store_vec(const StageRegs<DT> &r, const StageRows &sr, int d, int k0, float *sA, float *sB) {
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
  const bool k_live = k0 + q * 4 < d;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    f32x4 av;
    if constexpr (DT == MSAE_F32) av = r.a[it];
    else if constexpr (DT == MSAE_BF16) av = f32x4{bf16_bits_to_f32(r.a[it][0]), bf16_bits_to_f32(r.a[it][1]), bf16_bits_to_f32(r.a[it][2]), bf16_bits_to_f32(r.a[it][3])};
    else av = f32x4{f16_bits_to_f32(r.a[it][0]), f16_bits_to_f32(r.a[it][1]), f16_bits_to_f32(r.a[it][2]), f16_bits_to_f32(r.a[it][3])};
    av = av - r.bd;                                          // (a32 = f32(x) - b_dec, sae.py:174)
    f32x4 bv = r.b[it];
    const bool la = k_live && ((sr.live_a >> it) & 1u), lb = k_live && ((sr.live_b >> it) & 1u);
    // k = 4 q + e  ->  position f_kpos: e = 0, 2 at [2 q, 2 q + 1], e = 1, 3 at 16 + [2 q, 2 q + 1]
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float *pa = sA + row * F_PITCH + q * 2;
    float *pb = sB + row * F_PITCH + q * 2;
    *reinterpret_cast<f32x2 *>(pa) = la ? f32x2{av[0], av[2]} : f32x2{0.f, 0.f};
    *reinterpret_cast<f32x2 *>(pa + 16) = la ? f32x2{av[1], av[3]} : f32x2{0.f, 0.f};
    *reinterpret_cast<f32x2 *>(pb) = lb ? f32x2{bv[0], bv[2]} : f32x2{0.f, 0.f};
    *reinterpret_cast<f32x2 *>(pb + 16) = lb ? f32x2{bv[1], bv[3]} : f32x2{0.f, 0.f};
  }
}

// generic (unaligned / d % 4 != 0) path: element loads with bounds checks, converted at the load as before
struct StageRegsG {
  f32x4 a[4];
  f32x4 b[4];
};
template <int DT>
__device__ __forceinline__ void stage_load_gen(StageRegsG &r, const void *x, const float *W,
                                               const float *b_dec, const int *rows, int T, int d, int N,
                                               int m0, int n0, int k0) {
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
  const int kq = k0 + q * 4;
  f32x4 bd = {0.f, 0.f, 0.f, 0.f};
  if (b_dec) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bd[e] = (kq + e < d) ? b_dec[kq + e] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
    const int t = m0 + row;
    if (t < T) {
      const size_t xr = rows ? (size_t)rows[t] : (size_t)t;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (kq + e < d) av[e] = load_x1<DT>(x, xr * d + kq + e) - bd[e];
    }
    const int n = n0 + row;
    if (n < N) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (kq + e < d) bv[e] = W[(size_t)n * d + kq + e];
    }
    r.a[it] = av;
    r.b[it] = bv;
  }
}

__device__ __forceinline__ void stage_store_gen(const StageRegsG &r, float *sA, float *sB) {
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + it * 32;
    float *pa = sA + row * F_PITCH, *pb = sB + row * F_PITCH;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[f_kpos(q * 4 + e)] = r.a[it][e];
      pb[f_kpos(q * 4 + e)] = r.b[it][e];
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
  typename std::conditional<VEC, StageRegs<DT>, StageRegsG>::type regs;
  [[maybe_unused]] StageRows srows;
  if constexpr (VEC) srows = stage_rows<DT>(rows, T, d, N, m0, n0);
  auto load = [&](int k0) {
    if constexpr (VEC) stage_load_vec<DT>(regs, x, W, b_dec, srows, d, k0);
    else stage_load_gen<DT>(regs, x, W, b_dec, rows, T, d, N, m0, n0, k0);
  };
  auto store = [&](int k0, float *sA, float *sB) {
    if constexpr (VEC) stage_store_vec<DT>(regs, srows, d, k0, sA, sB);
    else stage_store_gen(regs, sA, sB);
  };
  load(0);
  store(0, smem, smem + F_BM * F_PITCH);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load((kt + 1) * F_BK);
    const float *a_base = smem + cur * STAGE + (wr * 64 + l31) * F_PITCH + khalf * 16;
    const float *b_base = smem + cur * STAGE + F_BM * F_PITCH + (wc * 64 + l31) * F_PITCH + khalf * 16;
    // k-step s of the tile multiplies k = 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63): position s of the lane's half.  Four steps
    // per 16-B read; the chain per output stays ascending in k (s = 0 .. 15 in order).
    f32x4 a0v = *reinterpret_cast<const f32x4 *>(a_base), a1v = *reinterpret_cast<const f32x4 *>(a_base + 32 * F_PITCH);
    f32x4 b0v = *reinterpret_cast<const f32x4 *>(b_base), b1v = *reinterpret_cast<const f32x4 *>(b_base + 32 * F_PITCH);
#pragma unroll
    for (int g = 0; g < F_BK / 8; ++g) {
      f32x4 a0n = a0v, a1n = a1v, b0n = b0v, b1n = b1v;
      if (g + 1 < F_BK / 8) {                                 // the next four steps' operands fly behind this group's 16 MFMAs
        a0n = *reinterpret_cast<const f32x4 *>(a_base + 4 * (g + 1)); a1n = *reinterpret_cast<const f32x4 *>(a_base + 32 * F_PITCH + 4 * (g + 1));
        b0n = *reinterpret_cast<const f32x4 *>(b_base + 4 * (g + 1)); b1n = *reinterpret_cast<const f32x4 *>(b_base + 32 * F_PITCH + 4 * (g + 1));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[e], b0v[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[e], b1v[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[e], b0v[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[e], b1v[e], acc[1][1], 0, 0, 0);
      }
      a0v = a0n; a1v = a1n; b0v = b0n; b1v = b1n;
    }
    if (kt + 1 < nk) store((kt + 1) * F_BK, smem + (cur ^ 1) * STAGE, smem + (cur ^ 1) * STAGE + F_BM * F_PITCH);
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
