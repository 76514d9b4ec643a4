// gemm_mfma64.h -- the candidate-pass GEMM of gemm_mfma.h with a DEEPER operand pipeline: 64-byte k-tiles in a 4-slot
// ring (three k-tiles = 96 KB in flight per CU) instead of 128-byte k-tiles in a 2-slot ring (one k-tile in flight).
//
// Why (round 3, tools/dma_depth.hip, profiles/r03_dma_tile_major.txt).  The 2 x 128 B ring is a latency chain: a k-tile is
// issued behind barrier i and must have landed before barrier i + 1, and with every CU streaming the L2 -> LDS delivery of
// one 64-KB k-tile takes ~1.5 us (43 GB/s per CU) -- as long as the k-tile's MFMA work (1.1-1.24 us at the clock the chip
// sustains), so the loop runs at the delivery LATENCY.  With three k-tiles in flight the stream is throughput-bound: 50.6
// GB/s per CU from row-major operands, 65.9 GB/s (1.00 us per 64 KB) from TILE-MAJOR operands whose 1-KiB staging pieces
// are contiguous kilobytes.  Round 2 built this ring on row-major operands and measured -5.5 %; the tile-major layout is
// what makes the stream fast enough to pay for the second barrier per 128 bytes of k.
//
// Layout.  Operands are tile-major for 64-byte k-tiles: [row tile of 256][k-tile of 64 B][row in tile][64 B], the four
// 16-B chunks of a row permuted by swz64(r) = (r >> 2) & 3 (position p holds chunk p ^ swz64(r)): every ds_read_b128
// fragment read of a 32-row block is bank-conflict-free (the four rows of one residue mod 4 inside each 16-lane service
// group land on four different chunk positions), and the LDS-DMA source is lane-linear (lane * 16).
//
// Protocol.  All k-tiles of all output tiles of a (persistent) workgroup form one flat sequence of POSITIONS; position p
// lives in ring slot p & 3.  Iteration p:   W  s_waitcnt vmcnt(pieces of p+1 and p+2): this wave's pieces of p landed
//                                           B  s_barrier: everybody's landed; everybody is done reading slot (p-1) & 3
//                                           S  stage position p + 3 into slot (p + 3) & 3 = (p - 1) & 3
//                                           C  fragment reads + MFMAs of position p
// Loads issued later than p + 2 (the next tile's epilogue constants) only make W wait longer, never shorter.  The
// outlier tile is position 0 of every output tile (compact: 32 B per row, 2 pieces per wave) or positions 0 and 1 (more
// than 32 outlier dims: the two 64-byte halves of the [rows][128 B] outlier operands).
#pragma once
#include <type_traits>

#include "gemm_mfma.h"

template <int BM_, int BN_, int WM_, int WN_, bool I8_>
struct GemmCfg64 {
  static constexpr int BM = BM_, BN = BN_, STAGES = 4, WM = WM_, WN = WN_;
  static constexpr bool I8 = I8_;
  static constexpr bool ABL_NOSTAGE = false, ABL_NOREAD = false, ABL_NOMFMA = false;
  static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
  static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  static constexpr int ROWB = 64, KS = 2;
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_RING_BYTES = STAGES * STAGE_BYTES;
  static constexpr int SIDE_SLOTS = 6;
  static constexpr int SIDE_BYTES = SIDE_SLOTS * NT * 4;
  static constexpr int QCAP = 2304;
  static constexpr int PF_SINK_BYTES = 256, GRP_BYTES = 256;
  static constexpr int LDS_BYTES = LDS_RING_BYTES + SIDE_BYTES + 16 + QCAP * 8 + PF_SINK_BYTES + GRP_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVES, A_PIECES = A_BYTES / 1024;
  static_assert(BM == 256 && BN == 256 && NWAVES == 8 && PPW == 4, "written for 256 x 256 tiles and 8 waves");
  static_assert(BM + BN <= NT && LDS_BYTES <= 160 * 1024, "epilogue constants / LDS budget");
};

// byte offset of the 16-B chunk at column c (bytes, c % 16 == 0) of row r in a tile-major operand of 64-byte k-tiles
__host__ __device__ __forceinline__ size_t packed64_off(size_t r, int c, int d) {
  const size_t rt = r >> 8, ri = r & 255;
  const int kt = c >> 6, ch = (c >> 4) & 3;
  return ((rt * (size_t)(d >> 6) + kt) * 256 + ri) * 64 + (size_t)((ch ^ (int)((ri >> 2) & 3)) << 4);
}

template <int N>
__device__ __forceinline__ void gemm64_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 64-byte k-tile (two k-steps) of MFMAs out of ring slot image `sA`
template <class C>
__device__ __forceinline__ void gemm64_compute(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA, int wr, int wc,
                                               int l31, int kh) {
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)sA;
  const unsigned rowA = base + (unsigned)(wr * C::TM + l31) * 64u;
  const unsigned rowB = base + (unsigned)C::A_BYTES + (unsigned)(wc * C::TN + l31) * 64u;
  const unsigned sw = (unsigned)(l31 >> 2) & 3u;          // the same for every 32-row block of A and B
  const unsigned off0 = (((unsigned)kh) ^ sw) << 4, off1 = (((unsigned)(2 + kh)) ^ sw) << 4;
  i32x4 a0[C::MI], b0[C::NI], a1[C::MI], b1[C::NI];
  a0[0] = lds_read_b128<0 * 2048>(rowA + off0); a0[1] = lds_read_b128<1 * 2048>(rowA + off0);
  a0[2] = lds_read_b128<2 * 2048>(rowA + off0); a0[3] = lds_read_b128<3 * 2048>(rowA + off0);
  b0[0] = lds_read_b128<0 * 2048>(rowB + off0); b0[1] = lds_read_b128<1 * 2048>(rowB + off0);
  a1[0] = lds_read_b128<0 * 2048>(rowA + off1); a1[1] = lds_read_b128<1 * 2048>(rowA + off1);
  a1[2] = lds_read_b128<2 * 2048>(rowA + off1); a1[3] = lds_read_b128<3 * 2048>(rowA + off1);
  b1[0] = lds_read_b128<0 * 2048>(rowB + off1); b1[1] = lds_read_b128<1 * 2048>(rowB + off1);
  lgkm_wait_tied<6, C>(a0, b0);
  gemm_mfma_step<C>(acc, a0, b0);
  lgkm_wait_tied<0, C>(a1, b1);
  gemm_mfma_step<C>(acc, a1, b1);
}

template <class C, bool DENSE>
__global__ __launch_bounds__(C::NT) void gemm64_kernel(GemmOperands op, int T, int Tp, int N, int nM, int nN,
                                                       GemmEpilogue ep) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int n_out_tiles = nM * nN;
  const bool has_out = C::I8 && op.Ao != nullptr;
  int lead_ks = 0;
  if (has_out) lead_ks = op.n_out ? (*op.n_out + 31) >> 5 : 4;       // wave-uniform scalar load
  const bool lead_compact = lead_ks <= 1;
  const int nlead = has_out ? (lead_compact ? 1 : 2) : 0;
  const int ntiles = nlead + op.nk;                                  // positions per output tile (op.nk: 64-byte k-tiles)
  const int krot = (C::I8 && nM == 1) ? (int)(gridDim.x >= 64 ? blockIdx.x >> 3 : blockIdx.x) % op.nk : 0;

  // ---- staging cursor: the next position to stage = (output tile s_id, k index s_kt); c1 / c2 = pieces this wave has in
  // flight for the positions one and two ahead of the one being computed
  int s_id = blockIdx.x, s_kt = 0, s_m0 = 0, s_n0 = 0, s_seq = 0;
  {
    int tm, tn;
    if (s_id < n_out_tiles) { gemm_map_tile(s_id, nM, nN, tm, tn); s_m0 = tm * C::BM; s_n0 = tn * C::BN; }
  }
  auto stage_next = [&](int wave, int lane) -> int {                 // -> pieces issued by this wave (0: nothing left)
    if (s_id >= n_out_tiles) return 0;
    unsigned char *base = smem + (s_seq & 3) * C::STAGE_BYTES;
    int issued;
    if (s_kt < nlead) {
      if (lead_compact) {
        gemm_stage_lead_compact<C>(op.Ao, op.Bo, s_m0, s_n0, smem, s_seq & 3, wave, lane);
        issued = 2;
      } else {                                                       // half s_kt of the [rows][128 B] outlier operands
        const unsigned voff = (unsigned)(lane >> 2) * 128u + (unsigned)s_kt * 64u +
                              ((((unsigned)lane & 3u) ^ (((unsigned)lane >> 4) & 3u)) << 4);
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
          const int piece = wave * C::PPW + i;
          const bool isA = piece < C::A_PIECES;
          const int pl = isA ? piece : piece - C::A_PIECES;
          const unsigned char *sbase = (isA ? op.Ao + (size_t)s_m0 * 128 : op.Bo + (size_t)s_n0 * 128) + (size_t)pl * 16 * 128;
          const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(base + piece * 1024);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                       :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
        }
        issued = C::PPW;
      }
    } else {
      int kq = s_kt - nlead + krot;
      kq -= kq >= op.nk ? op.nk : 0;
      const unsigned char *tileA = op.A + ((size_t)(s_m0 / C::BM) * op.nk + kq) * C::A_BYTES;
      const unsigned char *tileB = op.B + ((size_t)(s_n0 / C::BN) * op.nk + kq) * C::B_BYTES;
      const unsigned voff = (unsigned)lane << 4;
#pragma unroll
      for (int i = 0; i < C::PPW; ++i) {
        const int piece = wave * C::PPW + i;
        const bool isA = piece < C::A_PIECES;
        const unsigned char *sbase = isA ? tileA + piece * 1024 : tileB + (piece - C::A_PIECES) * 1024;
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(base + piece * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
      }
      issued = C::PPW;
    }
    ++s_seq;
    if (++s_kt == ntiles) {
      s_kt = 0;
      s_id += gridDim.x;
      if (s_id < n_out_tiles) {
        int tm, tn;
        gemm_map_tile(s_id, nM, nN, tm, tn);
        s_m0 = tm * C::BM; s_n0 = tn * C::BN;
      }
    }
    return issued;
  };

  int seq = 0;                                           // position being computed; its slot is seq & 3
  int c0 = 0, c1 = 0, c2 = 0;                            // pieces in flight for positions seq, seq + 1, seq + 2
  {
    const int lane0 = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c0 = stage_next(wave0, lane0);
    c1 = stage_next(wave0, lane0);
    c2 = stage_next(wave0, lane0);
  }
  (void)c0;
  for (int tile_id = blockIdx.x; tile_id < n_out_tiles; tile_id += gridDim.x) {
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                        // (see gemm_kernel: keeps per-tile addresses out of the k-loop)
    const int lane = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int l31 = lane & 31, kh = lane >> 5;
    int tm, tn;
    gemm_map_tile(tile_id, nM, nN, tm, tn);
    const int m0 = tm * C::BM, n0 = tn * C::BN;

    // epilogue constants of this tile: fetched now into registers, parked in LDS after the k-loop
    float side0 = 0.f, side1 = 0.f, side3 = 0.f, side4 = 0.f;
    int side2 = 1;
    float ref0 = 1.f, ref1 = 1.f, ref2 = 1.f;
    {
      if constexpr (!DENSE) { ref0 = ep.refs[0]; ref1 = ep.refs[1]; ref2 = ep.refs[2]; }
      const int tid = tid_;
      if (tid < C::BM) {
        const int t = m0 + tid;
        if constexpr (!DENSE) {
          const float v = (t < T) ? ep.tau_vals[(size_t)t * ep.tau_ld + ep.tau_col] : 0.f;
          side0 = (v > 0.f) ? v : __builtin_inff();
        }
        if (t < T) {
          const f32x4 rc = ep.rowc[t];
          side1 = rc[0];
          side3 = rc[2];
          side4 = 1.f;
          if (has_out) { side2 = (int)rc[1]; side4 = rc[1]; }
        }
      } else if (tid < C::BM + C::BN) {
        const int n = n0 + tid - C::BM;
        const int feat = n * ep.bias_stride + ep.bias_off;
        side0 = ep.bias ? ep.bias[feat] : 0.f;
        const f32x4 cc = ep.colc[n];
        side1 = cc[0];
        side2 = __float_as_int(cc[1]);
        side3 = cc[2];
        side4 = cc[3];
      }
    }

    f32x16 acc[C::MI][C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
      for (int j = 0; j < C::NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int *side_m = reinterpret_cast<int *>(smem + C::LDS_RING_BYTES) + 2 * C::NT;
    // one position: W (wait) - B (barrier) - S (stage position + 3) - C (compute); MODE 0 = main k-tile, 1 = compact
    // outlier tile, 2 = one 64-byte half of the full outlier tile
    auto iteration = [&](auto mode_tag, bool park_m) {
      constexpr int MODE = decltype(mode_tag)::value;
#if defined(MSAE_R64_ABL) && (MSAE_R64_ABL & 1)             // tuning: drain everything every iteration
      gemm64_wait<0>();
#else
      switch (c1 + c2) {                                 // W: at most the pieces of the two younger positions outstanding
        case 8: gemm64_wait<8>(); break;
        case 6: gemm64_wait<6>(); break;
        case 4: gemm64_wait<4>(); break;
        case 2: gemm64_wait<2>(); break;
        default: gemm64_wait<0>(); break;
      }
#endif
      if (park_m) {                                      // outlier multipliers of the tile's rows -> LDS
        if (tid_ < C::BM) side_m[tid_] = side2;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                      // B
#if defined(MSAE_R64_ABL) && (MSAE_R64_ABL & 2)             // tuning: no staging in the loop (results invalid)
      const int c3 = 0;
#else
      const int c3 = stage_next(wave, lane);             // S: position seq + 3 -> slot (seq + 3) & 3
#endif
      const unsigned char *sA = smem + (seq & 3) * C::STAGE_BYTES;
      if constexpr (MODE == 1) {
        if (lead_ks > 0) { if constexpr (C::I8) gemm_compute_lead_compact<C>(acc, sA, wr, wc, l31, kh); }
      } else {
#if !(defined(MSAE_R64_ABL) && (MSAE_R64_ABL & 4))          // tuning: no fragment reads / MFMAs (results invalid)
        gemm64_compute<C>(acc, sA, wr, wc, l31, kh);
#endif
      }
      ++seq;
      c1 = c2; c2 = c3;
    };
    int kt = 0;
    if (has_out) {                                       // peeled: the outlier dims were quantised at scale m[t] * sx[t]
      if (lead_compact) {
        iteration(std::integral_constant<int, 1>(), true);
        kt = 1;
      } else {
        iteration(std::integral_constant<int, 2>(), true);
        iteration(std::integral_constant<int, 2>(), false);
        kt = 2;
      }
      if (lead_ks > 0) {                                 // acc *= m[t] (exact 24-bit multiply, see gemm_kernel)
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = side_m[wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
              i32x16 v = __builtin_bit_cast(i32x16, acc[i][j]);
              v[e] = __mul24(v[e], m);
              acc[i][j] = __builtin_bit_cast(f32x16, v);
            }
          }
      }
    }
    for (; kt < ntiles; ++kt) iteration(std::integral_constant<int, 0>(), false);

    // park the epilogue constants (as gemm_kernel)
    float *side = reinterpret_cast<float *>(smem + C::LDS_RING_BYTES);
    side[tid_] = side0;
    side[C::NT + tid_] = side1;
    reinterpret_cast<int *>(side)[2 * C::NT + tid_] = side2;
    side[3 * C::NT + tid_] = side3;
    side[4 * C::NT + tid_] = side4;
    if constexpr (!DENSE) {
      float side5 = 0.f;
      if (tid_ < C::BM) {
        float b2 = side3 * ref0;
        if constexpr (C::I8) {
          const float rz = side1 * side1 * ep.zz12;
          b2 = __builtin_fmaf(rz * side4 * side4, ref2, __builtin_fmaf(rz, ref1, b2));
        }
        side5 = __builtin_sqrtf(b2) * 1.00001f;
      } else if (tid_ < C::BM + C::BN) {
        const float q = __int_as_float(side2);
        float h2 = q / ref0;
        if constexpr (C::I8) {
          h2 = fmaxf(h2, side3 / ref1);
          if (side4 > 0.f) h2 = fmaxf(h2, side4 / ref2);
        }
        side5 = (q > 0.f || side3 > 0.f || side4 > 0.f) ? __builtin_sqrtf(h2) * 1.00001f : 0.f;
      }
      side[5 * C::NT + tid_] = side5;
    }
    gemm_epilogue<C, DENSE>(acc, ep, T, m0, n0, wr, wc, lane, smem, side, [] {}, 0);
  }
  gemm64_wait<0>();
}

// Host launcher: operands tile-major for 64-byte k-tiles (op.packed), op.nk = d / 64.
template <class C, bool DENSE>
inline int gemm64_launch(const GemmOperands &op, int T, int Tp, int N, const GemmEpilogue &ep, hipStream_t s) {
  if (Tp % C::BM || N % C::BN || op.nk < 3 || !op.packed) return MSAE_EINVAL;
  const int nM = Tp / C::BM, nN = N / C::BN;
  auto kern = gemm64_kernel<C, DENSE>;
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  static int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 8 ? cus / 8 * 8 : 8;
  }();
  const int grid = nM * nN <= n_cu ? nM * nN : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, op, T, Tp, N, nM, nN, ep);
  return (int)hipGetLastError();
}
