// gemm_bf16.h -- C[T][N] = A[T][d] * B[N][d]^T on v_mfma_f32_32x32x16_bf16, with the SAE epilogues.
//
// The dominant kernel of the fused encoder (encode_fused.hip).  Roofline: bf16 MFMA (2.5 PFLOP/s
// dense on gfx950), 2*d*N FLOP per token.  Both operands are K-contiguous bf16 (A = bf16(x - b_dec),
// B = bf16(W_enc)), so A and B fragments are read the same way.
//
// Structure (template parameters BM, BN, BK, STAGES, WM, WN):
//   * a workgroup of WM x WN waves computes a BM x BN tile; each wave owns (BM/WM) x (BN/WN) as
//     MI x NI blocks of 32x32 (16 accumulator VGPRs each);
//   * operand tiles travel HBM/L2 -> LDS by global_load_lds (16 B per lane, 1 KiB per wave
//     instruction, no VGPR round trip) into a ring of STAGES slots; the loop keeps STAGES-1 k-tiles
//     in flight and waits with a COUNTED s_waitcnt vmcnt(N) + one raw s_barrier per k-tile, so
//     loads stay in flight across barriers (an s_waitcnt vmcnt(0) per tile left the kernel bound
//     by L2/HBM latency: 781 TFLOP/s at 128x128x64 / 2 stages);
//   * LDS image: row r of a tile at byte r*ROWB with its 16-B chunks XOR-permuted by
//     swz(r) = (r / RPB) % CPR (RPB = rows per 256-B bank row, CPR = chunks per row), which makes
//     every ds_read_b128 fragment read bank-conflict-free.  global_load_lds writes lane-linear, so
//     the permutation is applied to the per-lane SOURCE address and again on the read;
//   * tile -> workgroup map is XCD-aware: the 32 workgroups resident on one XCD's 32 CUs form an
//     8 (M) x 4 (N) super-tile sharing 8 A-tiles and 4 B-tiles in that XCD's private L2.
//
// Epilogues: DENSE  out[t][n] = acc + bias[feature(n)]                      (sample pass)
//            THRESH append (feature, acc + bias) to token t's candidate list when > tau[t]
#pragma once
#include "common.h"

struct GemmEpilogue {
  const float *bias;             // b_enc
  int bias_stride, bias_off;     // feature of column n is n*bias_stride + bias_off
  float *dense; int ld_dense;    // DENSE
  const float *tau_vals; int tau_ld, tau_col;   // THRESH: tau[t] = tau_vals[t*tau_ld + tau_col]
  int *cnt; unsigned long long *cand; int cap;  // candidate lists
  int skip_a, skip_b;            // features never emitted (hook edits replace their latents)
};

// FLAGS: bit 0 = raise wave priority around each MFMA cluster (s_setprio); bit 1 = spread the next
// stage's LDS-DMA issue over the k-steps of the current one instead of issuing it up front.
// (An L2-prefetch variant -- one sparse dword load per wave per k-tile, 2-6 tiles ahead -- measured
// 954 vs 1180 TFLOP/s and was dropped: L2 miss latency is not what the loop waits for.)
template <int BM_, int BN_, int BK_, int STAGES_, int WM_, int WN_, int FLAGS_ = 0>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, STAGES = STAGES_, WM = WM_, WN = WN_;
  static constexpr bool PRIO = FLAGS_ & 1, SPREAD = FLAGS_ & 2;
  static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
  static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  static constexpr int ROWB = BK * 2;            // bytes per tile row
  static constexpr int CPR = ROWB / 16;          // 16-B chunks per row
  static constexpr int RPB = 256 / ROWB;         // rows per 256-B LDS bank row
  static constexpr int RPP = 1024 / ROWB;        // rows per 1-KiB glds piece
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVES;  // pieces per wave
  static constexpr int A_PIECES = A_BYTES / 1024;
  static_assert(PIECES % NWAVES == 0, "stage must split evenly over the waves");
  static_assert(TM % 32 == 0 && TN % 32 == 0 && BK % 16 == 0, "tile shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C>
__device__ __forceinline__ int gemm_swz(int row) { return (row / C::RPB) % C::CPR; }

// Issue this wave's share of k-tile `kt` into ring slot `slot`.
template <class C, int I0 = 0, int I1 = C::PPW>
__device__ __forceinline__ void gemm_stage(const unsigned short *__restrict__ A,
                                           const unsigned short *__restrict__ B, int m0, int n0,
                                           int Tp, int N, int d, int kt, unsigned char *lds, int slot,
                                           int wave, int lane) {
  unsigned char *base = lds + slot * C::STAGE_BYTES;
  const int r_in = lane / C::CPR, c_in = lane % C::CPR;
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const int piece = wave * C::PPW + i;              // wave-uniform
    const bool isA = piece < C::A_PIECES;
    const int pl = isA ? piece : piece - C::A_PIECES;  // piece index inside its operand tile
    const int r = pl * C::RPP + r_in;                 // tile row filled by this lane
    const int c = c_in ^ gemm_swz<C>(r);              // global chunk landing in LDS slot (r, c_in)
    int grow = (isA ? m0 : n0) + r;
    const int gmax = isA ? Tp : N;
    grow = grow < gmax ? grow : gmax - 1;             // rows past the end are loaded but never used
    const unsigned short *src = (isA ? A : B) + (size_t)grow * d + kt * C::BK + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(base + piece * 1024),
                                     16, 0, 0);
  }
}

template <class C>
__device__ __forceinline__ bf16x8 gemm_frag(const unsigned char *tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8 *>(tile + row * C::ROWB + ((chunk ^ gemm_swz<C>(row)) << 4));
}

// tile id -> (tm, tn); see header.  Falls back to M-fastest order when the grid does not factor.
__device__ __forceinline__ void gemm_map_tile(int b, int nM, int nN, int &tm, int &tn) {
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class C, bool DENSE>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[C::MI][C::NI], const GemmEpilogue &ep, int T,
                                              int m0, int n0, int wr, int wc, int lane) {
  const int l31 = lane & 31, kh = lane >> 5;
  // epilogue.  C[i][n] of a 32x32 block: n = lane&31, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if constexpr (DENSE) {
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int n = n0 + wc * C::TN + j * 32 + l31;
      const float bn = ep.bias ? ep.bias[n * ep.bias_stride + ep.bias_off] : 0.f;
#pragma unroll
      for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int t = m0 + wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
          if (t < T) ep.dense[(size_t)t * ep.ld_dense + n] = acc[i][j][e] + bn;
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
      // the 16 per-lane thresholds of this row block first (independent loads), then the compares
      float tau[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = m0 + wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const float v = (t < T) ? ep.tau_vals[(size_t)t * ep.tau_ld + ep.tau_col] : 0.f;
        tau[e] = (v > 0.f) ? v : __builtin_inff();  // degenerate / padded token: emit nothing
      }
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        const int feat = n0 + wc * C::TN + j * 32 + l31;
        const float bn = ep.bias ? ep.bias[feat] : 0.f;
        const bool live = (feat != ep.skip_a) && (feat != ep.skip_b);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[i][j][e] + bn;
          if (v > tau[e] && live) {
            const int t = m0 + wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const int slot = atomicAdd(ep.cnt + t, 1);
            if (slot < ep.cap)
              ep.cand[(size_t)t * ep.cap + slot] =
                  ((unsigned long long)f32_order_key(v) << 32) | (unsigned)(0x7FFFFFFF - feat);
          }
        }
      }
    }
  }
}

template <class C, bool DENSE>
__global__ __launch_bounds__(C::NT) void gemm_bf16_kernel(const unsigned short *__restrict__ A,
                                                          const unsigned short *__restrict__ B, int T,
                                                          int Tp, int d, int N, int nM, int nN,
                                                          GemmEpilogue ep) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / C::WN, wc = wave % C::WN;
  int tm, tn;
  gemm_map_tile(blockIdx.x, nM, nN, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < C::NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = d / C::BK;
#pragma unroll
  for (int s = 0; s < C::STAGES - 1; ++s)
    if (s < nk) gemm_stage<C>(A, B, m0, n0, Tp, N, d, s, smem, s, wave, lane);

  const int l31 = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    // k-tile kt has landed once at most STAGES-2 younger groups of this wave are outstanding
    if (kt + C::STAGES - 2 < nk) wait_vmcnt<C::PPW *(C::STAGES - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // all waves' pieces of kt landed; slot of kt-1 is free again
    const bool more = kt + C::STAGES - 1 < nk;
    const int nkt = kt + C::STAGES - 1, nslot = nkt % C::STAGES;
    if constexpr (!C::SPREAD) {
      if (more) gemm_stage<C>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
    }
    const unsigned char *sA = smem + (kt % C::STAGES) * C::STAGE_BYTES;
    const unsigned char *sB = sA + C::A_BYTES;
    constexpr int KS = C::BK / 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (C::SPREAD) {
        constexpr int per = (C::PPW + KS - 1) / KS;
        if (more) {
          if (ks == 0) gemm_stage<C, 0, (per < C::PPW ? per : C::PPW)>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
          if (ks == 1) gemm_stage<C, (per < C::PPW ? per : C::PPW), (2 * per < C::PPW ? 2 * per : C::PPW)>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
          if (ks == 2) gemm_stage<C, (2 * per < C::PPW ? 2 * per : C::PPW), (3 * per < C::PPW ? 3 * per : C::PPW)>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
          if (ks == 3) gemm_stage<C, (3 * per < C::PPW ? 3 * per : C::PPW), C::PPW>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
        }
      }
      const int chunk = ks * 2 + kh;
      bf16x8 a[C::MI], b[C::NI];
#pragma unroll
      for (int i = 0; i < C::MI; ++i) a[i] = gemm_frag<C>(sA, wr * C::TM + i * 32 + l31, chunk);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) b[j] = gemm_frag<C>(sB, wc * C::TN + j * 32 + l31, chunk);
      if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(0);
    }
  }

  gemm_epilogue<C, DENSE>(acc, ep, T, m0, n0, wr, wc, lane);
}

// ---- ping-pong schedule -------------------------------------------------------------------------
// Same tiles, LDS image and epilogues; different time structure.  The 8 waves form two groups of 4
// (one wave of each group per SIMD).  A k-tile is BK/16 steps; every step is a LOAD phase (6
// fragment ds_reads + this wave's share of the LDS-DMA for the tile STAGES-1 ahead) followed by a
// COMPUTE phase (MI*NI MFMAs under s_setprio 1), phases separated by s_barrier.  Group B runs one
// phase behind group A, so on every SIMD one wave is always in its MFMA phase while its partner
// reads LDS / issues DMA: the matrix pipe no longer drains at the per-tile barrier.  LDS-DMA for
// a tile is issued >= 2 tiles before its first read and retired by a counted vmcnt one phase
// (one barrier) before that read; the ring needs STAGES >= 4 slots.
template <class C, bool DENSE>
__global__ __launch_bounds__(C::NT) void gemm_bf16_pp_kernel(const unsigned short *__restrict__ A,
                                                             const unsigned short *__restrict__ B,
                                                             int T, int Tp, int d, int N, int nM,
                                                             int nN, GemmEpilogue ep) {
  static_assert(C::STAGES >= 4 && C::NWAVES == 8, "ping-pong needs a 4-slot ring and 8 waves");
  constexpr int KS = C::BK / 16;                       // steps per k-tile
  constexpr int PER = (C::PPW + KS - 1) / KS;          // LDS-DMA pieces issued per LOAD phase
  static_assert(PER * KS == C::PPW, "pieces must split evenly over the steps");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                            // 0: group A, 1: group B (one phase behind)
  const int wr = wave / C::WN, wc = wave % C::WN;
  int tm, tn;
  gemm_map_tile(blockIdx.x, nM, nN, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < C::NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = d / C::BK;
#pragma unroll
  for (int s = 0; s < C::STAGES - 1; ++s)
    gemm_stage<C>(A, B, m0, n0, Tp, N, d, s < nk ? s : nk - 1, smem, s, wave, lane);
  wait_vmcnt<C::PPW *(C::STAGES - 2)>();               // tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();           // group B starts one phase later
  __builtin_amdgcn_sched_barrier(0);

  const int l31 = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char *sA = smem + (kt % C::STAGES) * C::STAGE_BYTES;
    const unsigned char *sB = sA + C::A_BYTES;
    int nkt = kt + C::STAGES - 1;                       // tile whose DMA is issued during this tile
    const int nslot = nkt % C::STAGES;
    nkt = nkt < nk ? nkt : nk - 1;                      // past the end: harmless re-load keeps vmcnt uniform
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // ---------------- LOAD phase
      const int chunk = ks * 2 + kh;
      bf16x8 a[C::MI], b[C::NI];
#pragma unroll
      for (int i = 0; i < C::MI; ++i) a[i] = gemm_frag<C>(sA, wr * C::TM + i * 32 + l31, chunk);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) b[j] = gemm_frag<C>(sB, wc * C::TN + j * 32 + l31, chunk);
      if (ks == 0) gemm_stage<C, 0, PER>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
      if (ks == 1) gemm_stage<C, PER, (2 * PER < C::PPW ? 2 * PER : C::PPW)>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
      if (ks == 2) gemm_stage<C, (2 * PER < C::PPW ? 2 * PER : C::PPW), (3 * PER < C::PPW ? 3 * PER : C::PPW)>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
      if (ks == 3) gemm_stage<C, (3 * PER < C::PPW ? 3 * PER : C::PPW), C::PPW>(A, B, m0, n0, Tp, N, d, nkt, smem, nslot, wave, lane);
      // group B's last LOAD phase of the tile is the phase before group A first reads tile kt+1:
      // retire this wave's pieces of tile kt+1 (leave tiles kt+2 .. kt+STAGES-1 in flight)
      if (ks == KS - 1 && grp == 1) wait_vmcnt<C::PPW *(C::STAGES - 2)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own ds_reads done before the slot can be refilled
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- COMPUTE phase
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (ks == KS - 1 && grp == 0) wait_vmcnt<C::PPW *(C::STAGES - 2)>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();           // pair group B's extra initial barrier
  wait_vmcnt<0>();
  gemm_epilogue<C, DENSE>(acc, ep, T, m0, n0, wr, wc, lane);
}

// Host launcher.  Requires Tp % BM == 0, N % BN == 0, d % BK == 0 (checked by the caller's plan).
template <class C, bool DENSE, bool PINGPONG = false>
inline int gemm_bf16_launch(const unsigned short *A, const unsigned short *B, int T, int Tp, int d,
                            int N, const GemmEpilogue &ep, hipStream_t s) {
  if (Tp % C::BM || N % C::BN || d % C::BK) return MSAE_EINVAL;
  const int nM = Tp / C::BM, nN = N / C::BN;
  if constexpr (PINGPONG) {
    auto kern = gemm_bf16_pp_kernel<C, DENSE>;
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(nM * nN), dim3(C::NT), C::LDS_BYTES, s, A, B, T, Tp, d, N, nM, nN, ep);
  } else {
    auto kern = gemm_bf16_kernel<C, DENSE>;
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(nM * nN), dim3(C::NT), C::LDS_BYTES, s, A, B, T, Tp, d, N, nM, nN, ep);
  }
  return (int)hipGetLastError();
}
