// encode_fused.hip -- fused Sae.encode: bf16 MFMA candidate pass + exact f32 re-score + TopK.
//
// Replaces Sae.encode = select_topk(pre_acts(x)) (reference sae/sae.py:172-185) without ever
// writing the dense [T][N] latents (512 KiB/token at N = 131072) to HBM.
//
// Pipeline per call (all on one stream, no host synchronisation):
//   1. prep_x        xb[T][d] = bf16(x - b_dec)                           (HBM, tiny)
//   2. gemm<DENSE>   coarse pre-acts of a 1/16 strided SAMPLE of the features -> [T][S] f32
//   3. topk (r-th)   tau[t] = r-th largest sample value: expected ~16*r features of the full
//                    width exceed tau[t]
//   4. gemm<THRESH>  THE DOMINANT KERNEL.  [T][d] x [d][N] on v_mfma_f32_32x32x16_bf16, LDS tiles
//                    filled by global_load_lds (16 B/lane), XOR-swizzled, double-buffered;
//                    epilogue: +b_enc, compare with tau[t], append (feature, coarse) of the rare
//                    survivors to a per-token candidate list.  Roofline: bf16 MFMA, 2*d*N FLOP
//                    per token; HBM traffic is the weights once per 8 token tiles.
//   5. select_rescore per token: order candidates by coarse value, re-score the best k+extra with
//                    the exact ascending-k f32 fma chain over the f32 W_enc rows, take the
//                    canonical top-k, and verify the guard band
//                        v_k(exact) > max(best non-rescored coarse, tau) + eps_t.
//                    Tokens that fail (or overflowed / had tau <= 0) are flagged.
//   6. exact path    flagged tokens (normally none) are recomputed by encode_f32 + topk through a
//                    device-side row list; their results overwrite step 5's.
//
// Outputs are therefore bit-identical to msae_pre_acts_f32 + msae_topk_f32 whenever the guard
// band holds, and ARE that path's outputs when it does not.
#include "common.h"

int msae_pre_acts_launch(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                         const float *b_dec, const int *rows, const int *n_rows, int T, int d, int N,
                         int relu, float *out, int ld_out, hipStream_t s);
int msae_topk_launch(const float *latents, int T, int N, int k, int ld, const int *n_rows,
                     float *vals, int32_t *idx, hipStream_t s);

namespace {

constexpr int SAMPLE_STRIDE = 16, SAMPLE_OFF = 7;
constexpr int FB_MAX = 128;         // tokens the in-call exact fallback can absorb
constexpr int EXACT_T_MAX = 255;    // below this many tokens the exact path is used directly

// ---- prepared encoder ------------------------------------------------------------------------
struct Prepared {
  unsigned magic;
  int N, d, S;
  size_t off_wb, off_ws, bytes;
};
constexpr unsigned PREP_MAGIC = 0x4D534145u;  // "MSAE"

__host__ __device__ inline bool fast_shape_ok(int N, int d) {
  return N % (SAMPLE_STRIDE * 128) == 0 && d % 64 == 0 && N >= SAMPLE_STRIDE * 128;
}

inline Prepared make_prepared(int N, int d) {
  Prepared p{};
  p.magic = PREP_MAGIC;
  p.N = N; p.d = d;
  p.S = fast_shape_ok(N, d) ? N / SAMPLE_STRIDE : 0;
  p.off_wb = 256;
  p.off_ws = p.off_wb + (p.S ? msae_align_up((size_t)N * d * 2, 256) : 0);
  p.bytes = p.off_ws + msae_align_up((size_t)p.S * d * 2, 256);
  return p;
}

// W_bf16[n][c] = bf16(W[n][c]); sample row j = row j*16+7.  grid-stride over 8-element groups.
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

// xb[t][c] = bf16((float)x[t][c] - b_dec[c]) for t < T, zero rows up to Tp.
template <int DT>
__global__ __launch_bounds__(256) void prep_x_kernel(const void *__restrict__ x,
                                                     const float *__restrict__ b_dec, int T, int Tp,
                                                     int d, unsigned short *__restrict__ xb) {
  const size_t groups = (size_t)Tp * d / 4;
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
    const size_t e = g * 4;
    const size_t t = e / d, c = e % d;
    u16x4 o = {0, 0, 0, 0};
    if ((int)t < T) {
      f32x4 v = load_x4<DT>(x, e);
      if (b_dec) v = v - *reinterpret_cast<const f32x4 *>(b_dec + c);
      o[0] = f32_to_bf16_bits(v[0]); o[1] = f32_to_bf16_bits(v[1]);
      o[2] = f32_to_bf16_bits(v[2]); o[3] = f32_to_bf16_bits(v[3]);
    }
    *reinterpret_cast<u16x4 *>(xb + e) = o;
  }
}

// ---- bf16 MFMA GEMM ----------------------------------------------------------------------------
constexpr int G_BM = 128, G_BN = 128, G_BK = 64, G_THREADS = 256;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;           // 16 KiB per operand tile
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;         // A + B
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;          // double buffered: 64 KiB

// LDS image of a [128 rows][64 bf16] tile: row r at byte r*128, its eight 16-B chunks permuted
// by chunk' = chunk ^ ((r >> 1) & 7).  global_load_lds writes lane-linear (dest = base + lane*16),
// so the permutation is applied to the per-lane SOURCE address and again on the fragment read.
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void stage_tile(const unsigned short *__restrict__ g, int row0,
                                           int row_max, int ld, int k0, unsigned char *lds_tile,
                                           int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;            // 1 KiB = 8 rows of the tile
    const int r = piece * 8 + (lane >> 3);     // tile row this lane fills
    const int c = (lane & 7) ^ swz(r);         // global chunk that lands in LDS slot (r, lane&7)
    int grow = row0 + r;
    grow = grow < row_max ? grow : row_max - 1;  // clamp: rows past the end are never used
    const unsigned short *src = g + (size_t)grow * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)src,
        (__attribute__((address_space(3))) void *)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 read_frag(const unsigned char *lds_tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8 *>(lds_tile + row * 128 + ((chunk ^ swz(row)) << 4));
}

// tile id -> (tm, tn).  Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends
// on it).  Each XCD walks super-tiles of 8 (M) x 4 (N) tiles so the 32 concurrently resident
// workgroups of an XCD share 8 A-tiles and 4 B-tiles in that XCD's L2.
__device__ __forceinline__ void map_tile(int b, int nM, int nN, int &tm, int &tn) {
  constexpr int GM = 8, GN = 4;
  if (nM % GM == 0 && nN % GN == 0 && ((nM / GM) * (nN / GN)) % 8 == 0) {
    const int xcd = b & 7, slot = b >> 3;
    const int grp = slot / (GM * GN), w = slot % (GM * GN);
    const int st = grp * 8 + xcd;
    const int nSM = nM / GM;
    tm = (st % nSM) * GM + (w % GM);
    tn = (st / nSM) * GN + (w / GM);
  } else {
    tm = b % nM;
    tn = b / nM;
  }
}

struct GemmEpilogue {
  const float *bias;     // b_enc
  int bias_stride, bias_off;   // feature of column n is n*bias_stride + bias_off
  float *dense; int ld_dense;  // DENSE: out[t][n] = acc + bias
  const float *tau_vals; int tau_ld, tau_col;  // THRESH: tau[t] = tau_vals[t*tau_ld + tau_col]
  int *cnt; unsigned long long *cand; int cap;  // candidate lists
  int skip_a, skip_b;  // features never emitted (hook edits replace their latents)
};

template <bool DENSE>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_bf16_kernel(
    const unsigned short *__restrict__ A, const unsigned short *__restrict__ B, int T, int Tp, int d,
    int N, int nM, int nN, GemmEpilogue ep) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  map_tile(blockIdx.x, nM, nN, tm, tn);
  const int m0 = tm * G_BM, n0 = tn * G_BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = d / G_BK;
  stage_tile(A, m0, Tp, d, 0, smem, wave, lane);
  stage_tile(B, n0, N, d, 0, smem + G_TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int l31 = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char *cur = smem + (kt & 1) * G_STAGE_BYTES;
    unsigned char *nxt = smem + ((kt + 1) & 1) * G_STAGE_BYTES;
    if (kt + 1 < nk) {
      stage_tile(A, m0, Tp, d, (kt + 1) * G_BK, nxt, wave, lane);
      stage_tile(B, n0, N, d, (kt + 1) * G_BK, nxt + G_TILE_BYTES, wave, lane);
    }
    const unsigned char *sA = cur, *sB = cur + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < G_BK / 16; ++ks) {
      const int chunk = ks * 2 + kh;
      const bf16x8 a0 = read_frag(sA, wr * 64 + l31, chunk);
      const bf16x8 a1 = read_frag(sA, wr * 64 + 32 + l31, chunk);
      const bf16x8 b0 = read_frag(sB, wc * 64 + l31, chunk);
      const bf16x8 b1 = read_frag(sB, wc * 64 + 32 + l31, chunk);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue.  C[i][n]: n = lane&31, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if constexpr (DENSE) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + l31;
      const float bn = ep.bias ? ep.bias[n * ep.bias_stride + ep.bias_off] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int t = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
          if (t < T) ep.dense[(size_t)t * ep.ld_dense + n] = acc[i][j][e] + bn;
        }
    }
  } else {
    // all 32 per-lane thresholds first (independent loads), then the rare-survivor compares
    float tau[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = (t < T) ? ep.tau_vals[(size_t)t * ep.tau_ld + ep.tau_col] : 0.f;
        tau[i][e] = (v > 0.f) ? v : __builtin_inff();  // degenerate / padded token: emit nothing
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int feat = n0 + wc * 64 + j * 32 + l31;
      const float bn = ep.bias ? ep.bias[feat] : 0.f;
      const bool live = (feat != ep.skip_a) && (feat != ep.skip_b);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[i][j][e] + bn;
          if (v > tau[i][e] && live) {
            const int t = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const int slot = atomicAdd(ep.cnt + t, 1);
            if (slot < ep.cap)
              ep.cand[(size_t)t * ep.cap + slot] =
                  ((unsigned long long)f32_order_key(v) << 32) | (unsigned)(0x7FFFFFFF - feat);
          }
        }
    }
  }
}

// ---- candidate select + exact re-score ----------------------------------------------------------
struct RescoreArgs {
  const void *x; const float *W_enc, *b_enc, *b_dec;
  const float *tau_vals; int tau_ld, tau_col;
  const int *cnt; const unsigned long long *cand; int cap;
  int T, d, N, k, n_rescore;
  int set_feature; float set_value; int zero_feature;
  float *vals; int32_t *idx; int32_t *status;
  int *flagged; int *n_flagged;
};

// dynamic LDS: keys[cap] u64 | a[d] f32 | res[nrp] u64 | errs[n_rescore] f32 | maxerr f32
template <int DT>
__global__ __launch_bounds__(256) void select_rescore_kernel(RescoreArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  float *a = reinterpret_cast<float *>(smem + (size_t)p.cap * 8);
  const int nrp = next_pow2(p.n_rescore + 1);
  unsigned long long *res = reinterpret_cast<unsigned long long *>(smem + (size_t)p.cap * 8 + (size_t)p.d * 4);
  float *errs = reinterpret_cast<float *>(res + nrp);
  float &s_maxerr = errs[p.n_rescore];  // all LDS lives in the one dynamic array (16-B aligned base)

  const int t = blockIdx.x;
  const int cnt = p.cnt[t];
  const int n = cnt < p.cap ? cnt : p.cap;
  const float tau = p.tau_vals[(size_t)t * p.tau_ld + p.tau_col];

  for (int i = threadIdx.x; i < p.cap; i += 256) keys[i] = (i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
  for (int c = threadIdx.x * 4; c < p.d; c += 1024) {
    f32x4 v = load_x4<DT>(p.x, (size_t)t * p.d + c);
    if (p.b_dec) v = v - *reinterpret_cast<const f32x4 *>(p.b_dec + c);
    *reinterpret_cast<f32x4 *>(a + c) = v;
  }
  for (int i = threadIdx.x; i < nrp; i += 256) res[i] = 0ull;
  if (threadIdx.x == 0) s_maxerr = 0.f;
  bitonic_sort_desc_u64(keys, p.cap);   // coarse value desc (index asc on ties); barriers inside

  // exact ascending-k chain for the best n_rescore coarse candidates: one lane per candidate
  const int C = n < p.n_rescore ? n : p.n_rescore;
  for (int c = threadIdx.x; c < C; c += 256) {
    const unsigned long long key = keys[c];
    const int f = rank_key_index(key);
    const float coarse = f32_from_order_key((unsigned)(key >> 32));
    const float *w = p.W_enc + (size_t)f * p.d;
    float acc = 0.f;
    for (int kk = 0; kk < p.d; kk += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + kk);
      const f32x4 av = *reinterpret_cast<const f32x4 *>(a + kk);
      acc = __builtin_fmaf(av[0], wv[0], acc);
      acc = __builtin_fmaf(av[1], wv[1], acc);
      acc = __builtin_fmaf(av[2], wv[2], acc);
      acc = __builtin_fmaf(av[3], wv[3], acc);
    }
    const float pre = acc + (p.b_enc ? p.b_enc[f] : 0.f);
    const float lat = pre > 0.f ? pre : 0.f;
    res[c] = rank_key(lat, f);
    errs[c] = fabsf(pre - coarse);
  }
  if (threadIdx.x == 0 && p.set_feature >= 0) res[p.n_rescore] = rank_key(p.set_value, p.set_feature);
  __syncthreads();
  // max |coarse - exact| over the re-scored candidates (wave 0)
  if (threadIdx.x < 64) {
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) m = fmaxf(m, errs[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if (threadIdx.x == 0) s_maxerr = m;
  }
  bitonic_sort_desc_u64(res, nrp);

  // verify the guard band and write
  const float eps = 4.f * s_maxerr + 1e-30f;
  const float v_k = f32_from_order_key((unsigned)(res[p.k - 1] >> 32));
  float bound = tau;
  if (n > C) bound = fmaxf(bound, f32_from_order_key((unsigned)(keys[C] >> 32)));
  const bool have_k = (C + (p.set_feature >= 0 ? 1 : 0)) >= p.k;
  const bool ok = (cnt <= p.cap) && (tau > 0.f) && have_k && (v_k > bound + eps);
  for (int j = threadIdx.x; j < p.k; j += 256) {
    const unsigned long long key = res[j];
    p.idx[(size_t)t * p.k + j] = key ? rank_key_index(key) : 0;
    p.vals[(size_t)t * p.k + j] = key ? f32_from_order_key((unsigned)(key >> 32)) : 0.f;
  }
  if (threadIdx.x == 0) {
    if (p.status) p.status[t] = ok ? 0 : 2;
    if (!ok) {
      const int slot = atomicAdd(p.n_flagged, 1);
      if (slot < FB_MAX) p.flagged[slot] = t;
    }
  }
}

__global__ void zero_i32_kernel(int *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

// hook edits on dense rows (exact path): latents[:, set_feature] = set_value; [:, zero_feature] = 0
__global__ void edit_dense_kernel(float *dense, int ld, int rows, const int *n_rows, int set_feature,
                                  float set_value, int zero_feature) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int R = n_rows ? min(rows, *n_rows) : rows;
  if (r >= R) return;
  if (set_feature >= 0) dense[(size_t)r * ld + set_feature] = set_value;
  if (zero_feature >= 0) dense[(size_t)r * ld + zero_feature] = 0.f;
}

// exact results of the flagged tokens overwrite the fast-path results
__global__ void scatter_fallback_kernel(const float *fb_vals, const int32_t *fb_idx, const int *flagged,
                                        const int *n_flagged, int k, float *vals, int32_t *idx,
                                        int32_t *status) {
  const int i = blockIdx.x;
  const int nf = min(*n_flagged, FB_MAX);
  if (i >= nf) return;
  const int t = flagged[i];
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    vals[(size_t)t * k + j] = fb_vals[(size_t)i * k + j];
    idx[(size_t)t * k + j] = fb_idx[(size_t)i * k + j];
  }
  if (threadIdx.x == 0 && status) status[t] = 1;
}

// ---- stage profiling (bench.py roofline): HIP events recorded on the launch stream ------------------
constexpr int PROF_MARKS = 7;  // boundaries of: prep | sample gemm | tau topk | main gemm | rescore | fallback
struct ProfState {
  bool on = false;
  int max_steps = 0, step = 0;
  hipEvent_t *ev = nullptr;
} g_prof;

inline void prof_mark(int i, hipStream_t s) {
  if (g_prof.on && g_prof.step < g_prof.max_steps)
    (void)hipEventRecord(g_prof.ev[g_prof.step * PROF_MARKS + i], s);
}

// ---- workspace carving -------------------------------------------------------------------------
struct FusedPlan {
  bool fast;
  int Tp, S, r, cap, n_rescore;
  size_t off_xb, off_sample, off_tauv, off_taui, off_cnt, off_cand, off_flag, off_fbdense, off_fbv,
      off_fbi, off_dense, bytes;
};

inline FusedPlan make_plan(int T, int d, int N, int k) {
  FusedPlan p{};
  p.fast = fast_shape_ok(N, d) && T > EXACT_T_MAX && k <= 256 && k >= 1;
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += msae_align_up(b, 256); return at; };
  if (p.fast) {
    p.Tp = (T + G_BM - 1) / G_BM * G_BM;
    p.S = N / SAMPLE_STRIDE;
    p.r = k / 4 > 16 ? k / 4 : 16;
    p.cap = next_pow2(64 * p.r);
    p.n_rescore = k + (k / 2 > 16 ? k / 2 : 16);
    p.off_xb = take((size_t)p.Tp * d * 2);
    p.off_sample = take((size_t)T * p.S * 4);
    p.off_tauv = take((size_t)T * p.r * 4);
    p.off_taui = take((size_t)T * p.r * 4);
    p.off_cnt = take((size_t)T * 4);
    p.off_cand = take((size_t)T * p.cap * 8);
    p.off_flag = take((size_t)(FB_MAX + 64) * 4);
    p.off_fbdense = take((size_t)FB_MAX * N * 4);
    p.off_fbv = take((size_t)FB_MAX * k * 4);
    p.off_fbi = take((size_t)FB_MAX * k * 4);
  } else {
    p.off_dense = take((size_t)T * N * 4);
  }
  p.bytes = o;
  return p;
}

template <int DT>
int run_fast(const void *x, const float *W_enc, const float *b_enc, const float *b_dec,
             const Prepared &pp, const unsigned char *prepared, int T, int d, int N, int k,
             int set_feature, float set_value, int zero_feature, float *vals, int32_t *idx,
             int32_t *status, unsigned char *ws, const FusedPlan &pl, hipStream_t s) {
  unsigned short *xb = reinterpret_cast<unsigned short *>(ws + pl.off_xb);
  float *sample = reinterpret_cast<float *>(ws + pl.off_sample);
  float *tauv = reinterpret_cast<float *>(ws + pl.off_tauv);
  int32_t *taui = reinterpret_cast<int32_t *>(ws + pl.off_taui);
  int *cnt = reinterpret_cast<int *>(ws + pl.off_cnt);
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + pl.off_cand);
  int *flagged = reinterpret_cast<int *>(ws + pl.off_flag);
  int *n_flagged = flagged + FB_MAX;
  float *fbdense = reinterpret_cast<float *>(ws + pl.off_fbdense);
  float *fbv = reinterpret_cast<float *>(ws + pl.off_fbv);
  int32_t *fbi = reinterpret_cast<int32_t *>(ws + pl.off_fbi);
  const unsigned short *wb = reinterpret_cast<const unsigned short *>(prepared + pp.off_wb);
  const unsigned short *wsamp = reinterpret_cast<const unsigned short *>(prepared + pp.off_ws);

  prof_mark(0, s);
  hipLaunchKernelGGL(zero_i32_kernel, dim3(64), dim3(256), 0, s, cnt, (size_t)T);
  hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, s, flagged, (size_t)(FB_MAX + 64));
  hipLaunchKernelGGL(prep_x_kernel<DT>, dim3(2048), dim3(256), 0, s, x, b_dec, T, pl.Tp, d, xb);

  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)gemm_bf16_kernel<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES));
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)gemm_bf16_kernel<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES));
  const int nM = pl.Tp / G_BM;
  prof_mark(1, s);
  {  // sample pass -> dense [T][S]
    GemmEpilogue ep{};
    ep.bias = b_enc; ep.bias_stride = SAMPLE_STRIDE; ep.bias_off = SAMPLE_OFF;
    ep.dense = sample; ep.ld_dense = pl.S;
    const int nN = pl.S / G_BN;
    hipLaunchKernelGGL(gemm_bf16_kernel<true>, dim3(nM * nN), dim3(G_THREADS), G_LDS_BYTES, s, xb,
                       wsamp, T, pl.Tp, d, pl.S, nM, nN, ep);
  }
  prof_mark(2, s);
  int rc = msae_topk_launch(sample, T, pl.S, pl.r, pl.S, nullptr, tauv, taui, s);
  if (rc) return rc;
  prof_mark(3, s);
  {  // full pass with the threshold epilogue
    GemmEpilogue ep{};
    ep.bias = b_enc; ep.bias_stride = 1; ep.bias_off = 0;
    ep.tau_vals = tauv; ep.tau_ld = pl.r; ep.tau_col = pl.r - 1;
    ep.cnt = cnt; ep.cand = cand; ep.cap = pl.cap;
    ep.skip_a = set_feature >= 0 ? set_feature : -1;
    ep.skip_b = zero_feature >= 0 ? zero_feature : -1;
    const int nN = N / G_BN;
    hipLaunchKernelGGL(gemm_bf16_kernel<false>, dim3(nM * nN), dim3(G_THREADS), G_LDS_BYTES, s, xb,
                       wb, T, pl.Tp, d, N, nM, nN, ep);
  }
  prof_mark(4, s);
  {
    RescoreArgs ra{};
    ra.x = x; ra.W_enc = W_enc; ra.b_enc = b_enc; ra.b_dec = b_dec;
    ra.tau_vals = tauv; ra.tau_ld = pl.r; ra.tau_col = pl.r - 1;
    ra.cnt = cnt; ra.cand = cand; ra.cap = pl.cap;
    ra.T = T; ra.d = d; ra.N = N; ra.k = k; ra.n_rescore = pl.n_rescore;
    ra.set_feature = set_feature; ra.set_value = set_value; ra.zero_feature = zero_feature;
    ra.vals = vals; ra.idx = idx; ra.status = status; ra.flagged = flagged; ra.n_flagged = n_flagged;
    const int nrp = next_pow2(pl.n_rescore + 1);
    const size_t smem = (size_t)pl.cap * 8 + (size_t)d * 4 + (size_t)nrp * 8 + (size_t)(pl.n_rescore + 4) * 4;
    auto kern = select_rescore_kernel<DT>;
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(T), dim3(256), smem, s, ra);
  }
  prof_mark(5, s);
  // exact recompute of flagged tokens (device-side count; empty grids exit immediately)
  rc = msae_pre_acts_launch(x, DT, W_enc, b_enc, b_dec, flagged, n_flagged, FB_MAX, d, N, 1, fbdense,
                            N, s);
  if (rc) return rc;
  if (set_feature >= 0 || zero_feature >= 0)
    hipLaunchKernelGGL(edit_dense_kernel, dim3(1), dim3(256), 0, s, fbdense, N, FB_MAX, n_flagged,
                       set_feature, set_value, zero_feature);
  rc = msae_topk_launch(fbdense, FB_MAX, N, k, N, n_flagged, fbv, fbi, s);
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_fallback_kernel, dim3(FB_MAX), dim3(64), 0, s, fbv, fbi, flagged,
                     n_flagged, k, vals, idx, status);
  prof_mark(6, s);
  if (g_prof.on && g_prof.step < g_prof.max_steps) ++g_prof.step;
  return msae_launch_status();
}

}  // namespace

extern "C" int msae_profile_begin(int max_steps) {
  if (max_steps <= 0 || max_steps > 4096) return MSAE_EINVAL;
  if (g_prof.ev) {
    for (int i = 0; i < g_prof.max_steps * PROF_MARKS; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    g_prof.ev = nullptr;
  }
  g_prof.ev = new hipEvent_t[(size_t)max_steps * PROF_MARKS];
  for (int i = 0; i < max_steps * PROF_MARKS; ++i) MSAE_HIP_TRY(hipEventCreate(&g_prof.ev[i]));
  g_prof.max_steps = max_steps;
  g_prof.step = 0;
  g_prof.on = true;
  return 0;
}

extern "C" int msae_profile_end(float *stage_ms, int *n_steps) {
  g_prof.on = false;
  const int n = g_prof.step;
  if (n_steps) *n_steps = n;
  for (int st = 0; st < n; ++st) {
    MSAE_HIP_TRY(hipEventSynchronize(g_prof.ev[st * PROF_MARKS + PROF_MARKS - 1]));
    for (int i = 0; i + 1 < PROF_MARKS; ++i)
      MSAE_HIP_TRY(hipEventElapsedTime(&stage_ms[st * (PROF_MARKS - 1) + i],
                                       g_prof.ev[st * PROF_MARKS + i], g_prof.ev[st * PROF_MARKS + i + 1]));
  }
  return 0;
}

extern "C" size_t msae_encoder_prepared_bytes(int N, int d) {
  if (N <= 0 || d <= 0) return 0;
  return make_prepared(N, d).bytes;
}

extern "C" int msae_encoder_prepare(const float *W_enc, int N, int d, void *prepared, void *stream) {
  if (N <= 0 || d <= 0 || !prepared) return MSAE_EINVAL;
  if (!msae_aligned(prepared, 256)) return MSAE_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  Prepared p = make_prepared(N, d);
  MSAE_HIP_TRY(hipMemcpyAsync(prepared, &p, sizeof(p), hipMemcpyHostToDevice, s));
  if (p.S) {
    if (!msae_aligned(W_enc, 16)) return MSAE_EALIGN;
    unsigned char *base = static_cast<unsigned char *>(prepared);
    hipLaunchKernelGGL(prepare_weights_kernel, dim3(4096), dim3(256), 0, s, W_enc, N, d,
                       reinterpret_cast<unsigned short *>(base + p.off_wb),
                       reinterpret_cast<unsigned short *>(base + p.off_ws));
  }
  return msae_launch_status();
}

extern "C" size_t msae_encode_topk_ws_bytes(int T, int d, int N, int k) {
  if (T <= 0 || d <= 0 || N <= 0 || k <= 0) return 0;
  return make_plan(T, d, N, k).bytes;
}

extern "C" int msae_encode_topk(const void *x, int x_dtype, const float *W_enc, const float *b_enc,
                                const float *b_dec, const void *prepared, int T, int d, int N, int k,
                                int set_feature, float set_value, int zero_feature, float *vals,
                                int32_t *idx, int32_t *status, void *ws, size_t ws_bytes,
                                void *stream) {
  if (T < 0 || d <= 0 || N <= 0 || k <= 0 || k > N || k > 4096) return MSAE_EINVAL;
  if (x_dtype != MSAE_F32 && x_dtype != MSAE_BF16 && x_dtype != MSAE_F16) return MSAE_EINVAL;
  if (set_feature >= N || zero_feature >= N) return MSAE_EINVAL;
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  FusedPlan pl = make_plan(T, d, N, k);
  if (!prepared && pl.fast) return MSAE_EINVAL;  // the fast path needs msae_encoder_prepare()
  if (ws_bytes < pl.bytes || !ws) return MSAE_EWS;
  if (!msae_aligned(ws, 256)) return MSAE_EALIGN;
  unsigned char *wsb = static_cast<unsigned char *>(ws);
  if (!pl.fast) {
    float *dense = reinterpret_cast<float *>(wsb + pl.off_dense);
    int rc = msae_pre_acts_launch(x, x_dtype, W_enc, b_enc, b_dec, nullptr, nullptr, T, d, N, 1,
                                  dense, N, s);
    if (rc) return rc;
    if (set_feature >= 0 || zero_feature >= 0)
      hipLaunchKernelGGL(edit_dense_kernel, dim3((T + 255) / 256), dim3(256), 0, s, dense, N, T,
                         (const int *)nullptr, set_feature, set_value, zero_feature);
    rc = msae_topk_launch(dense, T, N, k, N, nullptr, vals, idx, s);
    if (rc) return rc;
    if (status) hipLaunchKernelGGL(zero_i32_kernel, dim3(64), dim3(256), 0, s, status, (size_t)T);
    return msae_launch_status();
  }
  const Prepared pp = make_prepared(N, d);  // layout is a pure function of (N, d)
  const unsigned char *pb = static_cast<const unsigned char *>(prepared);
  if (!msae_aligned(x, x_dtype == MSAE_F32 ? 16 : 8) || !msae_aligned(W_enc, 16) ||
      (b_dec && !msae_aligned(b_dec, 16)))
    return MSAE_EALIGN;
  switch (x_dtype) {
    case MSAE_F32: return run_fast<MSAE_F32>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, s);
    case MSAE_BF16: return run_fast<MSAE_BF16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, s);
    default: return run_fast<MSAE_F16>(x, W_enc, b_enc, b_dec, pp, pb, T, d, N, k, set_feature, set_value, zero_feature, vals, idx, status, wsb, pl, s);
  }
}
