// gemm_skinny.h -- the candidate pass for ONE row of output tiles with few tokens (T <= 256): a weight STREAM.
//
// gemm_mfma.h keeps one 64-KB k-tile in flight per CU (2-slot LDS ring).  That is what a compute-bound 256 x 256 tile
// wants; a batch of 17 ... 128 tokens is bound by the HBM latency of Wq instead: its single row of output tiles
// streamed Wq at ~3 TB/s (0.18 ms of a 0.37 ms encode, T = 64).  Here the tile is BM tokens x 256 features and the
// operands take different routes:
//   B (Wq, HBM)   every wave owns 32 feature rows and loads ITS MFMA fragments straight from global memory -- 16 B
//                 per lane, 16 rows x 64 B per instruction (v_mfma_i32_16x16x64_i8's operand shape: 32 B per row, the
//                 32 x 32 x 32 shape, streamed at 3.3 TB/s) -- two batches of 2 x 4 k-steps in flight per wave
//                 (16 KB; 128 KB per CU ahead of the MFMAs, in registers, no LDS slot to wait for);
//   A (xq, L2)    the tokens' rows, 64 KB chunks of k through a double-buffered LDS image (every wave reads all of
//                 it): chunk c + 1 is fetched into registers while chunk c is consumed;
//   outlier tile  (int8 pass) its k-steps first, fragments of both operands from global, then acc *= m[t].
// The accumulators, the side constants and the epilogue are gemm_mfma.h's (gemm_epilogue: same candidate keys, same
// dense upper values), so both kernels are interchangeable per launch; int32 accumulation is exact in any order.
#pragma once
#include "gemm_mfma.h"

template <int BM_, int NW_ = 4>
struct SkinnyCfg {
  // NW waves per workgroup, 32 features each.  NW = 4: two workgroups share a CU (80 KB of LDS, 2 waves per SIMD at ~200
  // VGPRs each), so one's prologue / epilogue (~12 us each: dependent loads of the side constants, the outlier tile,
  // the queue flush's atomics) runs beside the other's stream; with ONE 8-wave workgroup per CU they were 50 us of a
  // 150 us pass.
  static constexpr int BM = BM_, NWAVES = NW_, BN = 32 * NW_, WM = 1, WN = NW_, NT = 64 * NW_;
  static constexpr int TM = BM, TN = 32, MI = BM / 32, NI = 1;
  static constexpr bool I8 = true, F8 = false, CERT = false, SCALED = true;   // (gemm_mfma.h's epilogue reads these)
  static constexpr int KC = (NW_ == 4 ? 32768 : 65536) / BM;   // bytes of k per A chunk
  // ADMA (the 256-token tile): the token rows travel L2 -> LDS by LDS-DMA (global_load_lds, as gemm_mfma.h's operands) instead of
  // through 32 registers per lane, which buys the B stream a second k-step in flight (UN = 2: 64 KB of weights in flight per CU
  // instead of 32).  DMA writes lane-linear, so the image is unpadded ([row][KC] with the row's sixteen 16-B chunks XOR-swizzled
  // by row & 15: the 16 rows of a fragment read land on 16 different bank groups).
  static constexpr bool ADMA = BM_ == 256 && NW_ == 8;
  static constexpr int PITCH = ADMA ? KC : KC + 16;     // row pitch of the LDS image (16 B pad: rows spread over the banks)
  static constexpr int A_BUF = BM * PITCH;
  static constexpr int LDS_RING_BYTES = 2 * A_BUF;
  static constexpr int SIDE_SLOTS = 7;                  // (slot 6: gemm_mfma.h's subtractive-dither corrections)
  static constexpr int SIDE_BYTES = SIDE_SLOTS * NT * 4;
  static constexpr int QCAP = 128 * NW_;                // ~0.5 % of BM x BN outputs pass the hot loop's bound
  static constexpr int LDS_BYTES = LDS_RING_BYTES + SIDE_BYTES + 16 + QCAP * 8;
  // k-steps (64 B of k each) per B batch (registers: 8 UN per batch; the accumulators take BM / 2 of the wave's budget)
  static constexpr int UN = ADMA ? MSAE_SK_UN / 2 : (BM_ > 128 ? MSAE_SK_UN / 4 : (BM_ > 64 ? MSAE_SK_UN / 2 : MSAE_SK_UN));
  static_assert(UN >= 1 && BM % 32 == 0 && BM + BN <= NT, "one thread per tile row and column fetches the epilogue constants");
  static_assert(KC % (2 * UN * 64) == 0, "a chunk is a whole number of B double-batches");
  static_assert(LDS_BYTES <= (NW_ == 4 ? 80 : 160) * 1024, "LDS budget");
  static_assert(!ADMA || KC == 256, "the DMA path's chunk swizzle covers sixteen 16-B chunks per row");
};

template <int BM, int NW, bool DENSE>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(GemmOperands op, int T, int d, GemmEpilogue ep) {
  using C = SkinnyCfg<BM, NW>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int n0 = blockIdx.x * C::BN;
  constexpr int UN = C::UN;
  // token tile blockIdx.y (the sample pass of 129 ... 256 tokens runs two 128-token tiles per feature block: twice the
  // workgroups on a launch that has few, and its 16 MB of weights are cheap to stream twice)
  const int m0 = blockIdx.y * C::BM;
  T = T - m0 < C::BM ? T - m0 : C::BM;                 // tokens of THIS tile
  op.A += (size_t)m0 * op.ldA;
  if (op.Ao) op.Ao += (size_t)m0 * 128;

  // ---- epilogue constants of this tile's rows / columns (registers now, LDS after the k-loop; as gemm_kernel)
  float side0 = 0.f, side1 = 0.f, side3 = 0.f, side4 = 0.f;
  int side2 = 1;
  int side6 = 0;                                       // subtractive dither (gemm_mfma.h): E_t (rows) | float bits of Ds_n (columns)
  float ref0 = 1.f, ref1 = 1.f, ref2 = 1.f;
  if constexpr (!DENSE) { ref0 = ep.refs[0]; ref1 = ep.refs[1]; ref2 = ep.refs[2]; }
  const bool has_out = op.Ao != nullptr;
  if (tid < C::BM) {
    const int t = tid;
    if constexpr (!DENSE) {
      const float v = (t < T) ? ep.tau_vals[(size_t)(m0 + t) * ep.tau_ld + ep.tau_col] : 0.f;
      side0 = (v > 0.f) ? v : __builtin_inff();       // degenerate / padded token: emit nothing
    }
    if (t < T) {
      const f32x4 rc = ep.rowc[m0 + t];
      side1 = rc[0];
      side3 = rc[2];
      side4 = 1.f;
      if (has_out) { side2 = (int)rc[1]; side4 = rc[1]; }
      if (ep.row_e) { const int2 em = ep.row_e[m0 + t]; side6 = em.x; side2 = em.y; }
    }
  } else if (tid < C::BM + C::BN) {
    const int n = n0 + tid - C::BM;
    const int feat = gemm_feature(ep, n);
    side0 = ep.bias ? ep.bias[feat] : 0.f;
    const f32x4 cc = ep.colc[n];
    side1 = cc[0];
    side2 = __float_as_int(cc[1]);
    side3 = cc[2];
    side4 = cc[3];
    if (ep.col_ds) side6 = __float_as_int(ep.col_ds[n]);
  }
  float *side = reinterpret_cast<float *>(smem + C::LDS_RING_BYTES);
  // (m, -E) of the rows as int2 in slot 2 (free until the constants are parked behind the k-loop): acc = acc * m - E below
  int2 *side_me = reinterpret_cast<int2 *>(reinterpret_cast<int *>(side) + 2 * C::NT);
  static_assert(2 * C::BM <= C::NT, "the (m, -E) pairs fit one side slot");
  if (tid < C::BM) side_me[tid] = int2{side2, -side6};
  const bool sub_e = ep.row_e != nullptr;              // (wave-uniform)

  // v_mfma_i32_16x16x64_i8: A = 16 tokens x 64 B of k, B = 16 features x 64 B of k (lane l: row l % 16, bytes 16 (l / 16) ..),
  // C[token 4 (l / 16) + r][feature l % 16] in 4 registers.  A wave owns 32 features = 2 feature groups, the tile's BM tokens
  // are BM / 16 token groups.
  constexpr int TG = C::BM / 16;
  const int l15 = lane & 15, lg = lane >> 4;
  i32x4 acc16[TG][2];
#pragma unroll
  for (int tg = 0; tg < TG; ++tg) { acc16[tg][0] = i32x4{0, 0, 0, 0}; acc16[tg][1] = i32x4{0, 0, 0, 0}; }

  // ---- A chunk 0 on its way (registers), then the outlier tile while it travels
  constexpr int A_PIECES = C::BM * C::KC / 16 / C::NT;   // 16-B pieces per thread and chunk (8)
  constexpr int CPR = C::KC / 16;                        // pieces per row
  i32x4 areg[A_PIECES];
  // only the row groups that hold tokens travel (piece q of a thread lies in rows q RPQ ... q RPQ + RPQ - 1); the others are
  // zeroed once in both buffers
  constexpr int RPQ = C::NT / CPR;                       // rows covered by one piece index q
  auto fetch_a = [&](int chunk) {
    if constexpr (C::ADMA) return;
#pragma unroll
    for (int q = 0; q < A_PIECES; ++q) {
      if (q * RPQ >= T) continue;                        // wave-uniform
      const int p = q * C::NT + tid, row = p / CPR, c16 = p % CPR;
      areg[q] = *reinterpret_cast<const i32x4 *>(op.A + (size_t)row * op.ldA + (size_t)chunk * C::KC + c16 * 16);
    }
  };
  auto put_a = [&](int buf) {
    if constexpr (C::ADMA) return;
#pragma unroll
    for (int q = 0; q < A_PIECES; ++q) {
      if (q * RPQ >= T) continue;
      const int p = q * C::NT + tid, row = p / CPR, c16 = p % CPR;
      *reinterpret_cast<i32x4 *>(smem + buf * C::A_BUF + row * C::PITCH + c16 * 16) = areg[q];
    }
  };
  // ADMA: one 1-KiB piece = 4 token rows x 256 B; wave w issues pieces 8 w .. 8 w + 7 of a chunk.  Lane l delivers chunk
  // (l & 15) ^ (row & 15) of row 4 piece + (l >> 4) to LDS position l & 15 of that row (the DMA writes lane-linear).  The rows of
  // the padded tile (>= T) are zero in xq, so every piece is always fetched.  4 per-lane offsets (piece & 3 decides row & 15).
  [[maybe_unused]] unsigned dma_off[4];
  if constexpr (C::ADMA) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned row_in = lane >> 4, pos = lane & 15, r15 = ((unsigned)j << 2) | row_in;
      dma_off[j] = row_in * (unsigned)op.ldA + ((pos ^ r15) << 4);
    }
  }
  auto dma_a = [&](int chunk, int buf) {
    if constexpr (C::ADMA) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int piece = __builtin_amdgcn_readfirstlane(wave) * 8 + q;   // wave-uniform: SGPR addresses below
        const unsigned char *sbase = op.A + (size_t)(piece * 4) * op.ldA + (size_t)chunk * C::KC;
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + buf * C::A_BUF + piece * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(dma_off[q & 3]), "s"(sbase), "s"(dst) : "memory", "m0");
      }
    }
  };
  if constexpr (!C::ADMA) {
#pragma unroll
    for (int q = 0; q < A_PIECES; ++q) {
      if (q * RPQ < T) continue;
      const int p = q * C::NT + tid, row = p / CPR, c16 = p % CPR;
      *reinterpret_cast<i32x4 *>(smem + row * C::PITCH + c16 * 16) = i32x4{0, 0, 0, 0};
      *reinterpret_cast<i32x4 *>(smem + C::A_BUF + row * C::PITCH + c16 * 16) = i32x4{0, 0, 0, 0};
    }
  }
  // k is walked in a ROTATED order: workgroup b starts at chunk b mod nchunks and every wave at its own batch inside the
  // chunk (all workgroups start together and advance in step; rows are d bytes apart: without the rotation every request
  // of the moment carries the same offset inside its row).
  const int nchunks = d / C::KC;
  constexpr int KSC = C::KC / 64;                       // 64-byte k-steps per chunk
  constexpr int BPC = KSC / (2 * UN);                   // B double-batches per chunk
  const int rot_c = blockIdx.x % nchunks;
  const int rot_b = BPC > 1 ? (wave + blockIdx.x / nchunks) % BPC : 0;
  auto chunk_of = [&](int c) { const int cc = c + rot_c; return cc >= nchunks ? cc - nchunks : cc; };
  auto ks_of = [&](int c, int b) { const int bq = b + rot_b; return chunk_of(c) * KSC + (bq >= BPC ? bq - BPC : bq) * 2 * UN; };
  fetch_a(chunk_of(0));
  dma_a(chunk_of(0), 0);

  // this lane's B fragment streams: feature groups 0 / 1 of the wave (rows 16 apart), 64 B of a row per k-step, from the
  // FRAGMENT-major Wq copy (encode_fused.hip: frag_off) -- [16-row block][k-step][lane][16 B]: the fragment of a k-step is one
  // contiguous kilobyte, lane l at byte 16 l, i.e. eight full 128-B lines per instruction.  (Fragments read from the row-major
  // copy -- 16 rows x 64 B per instruction -- are half-line requests: that stream ran at 3.3-4 TB/s whatever was in flight, and
  // the same from the tile-major copy, where the wave's rows of a k-tile are 4 KB of contiguous memory.  NOTEBOOK.md.)
  const unsigned char *brow0 = op.B + ((size_t)((n0 >> 4) + wave * 2) * (size_t)(d >> 6) << 10) + lane * 16;
  const unsigned char *brow1 = brow0 + ((size_t)(d >> 6) << 10);
  i32x4 ba[UN][2], bb[UN][2];
  auto load_b = [&](i32x4 (&dst)[UN][2], int ks) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      dst[u][0] = MSAE_SK_LOAD(reinterpret_cast<const i32x4 *>(brow0 + ((size_t)(ks + u) << 10)));
      dst[u][1] = MSAE_SK_LOAD(reinterpret_cast<const i32x4 *>(brow1 + ((size_t)(ks + u) << 10)));
    }
    MSAE_SK_FENCE();
  };
  load_b(ba, ks_of(0, 0));

  auto mfma = [&](int tg, const i32x4 &a, const i32x4 (&b)[2]) {
    acc16[tg][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[0], acc16[tg][0], 0, 0, 0);
    acc16[tg][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[1], acc16[tg][1], 0, 0, 0);
  };

  if (has_out) {
    const int lead_ks = op.n_out ? (*op.n_out + 63) >> 6 : 2;   // wave-uniform; the tile is zero beyond the dims present
    const unsigned char *bo = op.Bo + (size_t)(n0 + wave * 32 + l15) * 128 + lg * 16;
    for (int ks = 0; ks < lead_ks; ++ks) {
      const i32x4 b[2] = {*reinterpret_cast<const i32x4 *>(bo + ks * 64), *reinterpret_cast<const i32x4 *>(bo + 16 * 128 + ks * 64)};
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
        mfma(tg, *reinterpret_cast<const i32x4 *>(op.Ao + (size_t)(tg * 16 + l15) * 128 + ks * 64 + lg * 16), b);
    }
    __syncthreads();                                   // side_me written
    if (lead_ks > 0 || sub_e) {                        // |acc| < 2^21 and m <= 1040 (quant_x_kernel: larger multipliers send the token to the exact path): the product fits int32
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int2 me = side_me[tg * 16 + lg * 4 + r];
          acc16[tg][0][r] = __mul24(acc16[tg][0][r], me.x) + me.y;
          acc16[tg][1][r] = __mul24(acc16[tg][1][r], me.x) + me.y;
        }
    }
  }
  put_a(0);
  if constexpr (C::ADMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk 0 has landed (the compiler does not see the DMA)
  __syncthreads();

  // ---- main stream
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks && !(MSAE_SK_ABL & 1)) { fetch_a(chunk_of(c + 1)); dma_a(chunk_of(c + 1), (c + 1) & 1); }
    const unsigned char *abuf = smem + (c & 1) * C::A_BUF + l15 * C::PITCH + (C::ADMA ? 0 : lg * 16);
    // ADMA: the 16-B chunk (4 ks + lg) of row (16 tg + l15) sits at position chunk ^ l15
    auto a_frag = [&](int tg, int ksl) -> i32x4 {
      if constexpr (C::ADMA) return *reinterpret_cast<const i32x4 *>(abuf + tg * 16 * C::PITCH + ((((ksl << 2) | lg) ^ l15) << 4));
      else return *reinterpret_cast<const i32x4 *>(abuf + tg * 16 * C::PITCH + ksl * 64);
    };
#pragma unroll 1
    for (int b = 0; b < BPC; ++b) {
      const int ks = ks_of(c, b);                       // physical k-step of ba[0]
      const int k0 = ks - chunk_of(c) * KSC;            // ... inside the chunk
      if (!(MSAE_SK_ABL & 2)) load_b(bb, ks + UN);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
          if (!(MSAE_SK_ABL & 4)) mfma(tg, a_frag(tg, k0 + u), ba[u]);
          else asm volatile("" :: "v"(ba[u][0]), "v"(ba[u][1]));
        }
      }
      const bool last = c + 1 == nchunks && b + 1 == BPC;
      if (!last && !(MSAE_SK_ABL & 2)) load_b(ba, b + 1 < BPC ? ks_of(c, b + 1) : ks_of(c + 1, 0));
#pragma unroll
      for (int u = 0; u < UN; ++u) {
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
          if (!(MSAE_SK_ABL & 4)) mfma(tg, a_frag(tg, k0 + UN + u), bb[u]);
          else asm volatile("" :: "v"(bb[u][0]), "v"(bb[u][1]));
        }
      }
    }
    if (c + 1 < nchunks && !(MSAE_SK_ABL & 1)) {
      put_a((c + 1) & 1);                               // the other buffer: last read in iteration c - 1 (barrier below)
      // ADMA: the next chunk's pieces were issued at the top of this iteration; everything issued since is B loads, of which
      // only the last batch (2 UN loads: ba for the next iteration) may still be in flight -- vmcnt counts in order
      if constexpr (C::ADMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * UN) : "memory");
      // a raw barrier: __syncthreads() would also wait for vmcnt(0), i.e. drain the B batches in flight at every chunk
      // (measured: the stream ran at 3.3 TB/s with it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- accumulators -> the 32 x 32 block layout of gemm_epilogue (C[i][n]: n = lane & 31, i = (reg & 3) + 8 (reg >> 2) +
  // 4 (lane >> 5)) through an int32 image of the tile in the (now idle) A buffers
  f32x16 acc[C::MI][1];
  {
    constexpr int TP = C::BN + 4;                        // row pitch in ints: the 4 row groups of a store land 16 banks apart
    // the image of 128 token rows at a time (BM = 256: two halves through the same LDS)
    constexpr int HR = C::ADMA ? 64 : (C::BM < 128 ? C::BM : 128), NH = C::BM / HR;   // (the unpadded ADMA ring holds 64 image rows)
    static_assert(HR * TP * 4 <= C::LDS_RING_BYTES, "tile image fits the A buffers");
    int *timg = reinterpret_cast<int *>(smem);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      __syncthreads();                                   // every wave is done with the A buffers / the previous half
#pragma unroll
      for (int tg = h * (HR / 16); tg < (h + 1) * (HR / 16); ++tg)
#pragma unroll
        for (int fg = 0; fg < 2; ++fg)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            timg[((tg - h * (HR / 16)) * 16 + lg * 4 + r) * TP + wave * 32 + fg * 16 + l15] = acc16[tg][fg][r];
      __syncthreads();
#pragma unroll
      for (int i = h * (HR / 32); i < (h + 1) * (HR / 32); ++i) {
        i32x16 v;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          v[e] = timg[((i - h * (HR / 32)) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * TP + wave * 32 + l31];
        acc[i][0] = __builtin_bit_cast(f32x16, v);
      }
    }
  }

  // ---- epilogue constants -> LDS, then gemm_mfma.h's epilogue
  side[tid] = side0;
  side[C::NT + tid] = side1;
  reinterpret_cast<int *>(side)[2 * C::NT + tid] = side2;
  side[3 * C::NT + tid] = side3;
  side[4 * C::NT + tid] = side4;
  reinterpret_cast<int *>(side)[6 * C::NT + tid] = side6;
  if constexpr (!DENSE) {
    float side5 = 0.f;
    if (tid < C::BM) {          // B_t: z sigma of this token against the reference feature
      const float rz = side1 * side1 * ep.zz12;
      const float b2 = __builtin_fmaf(rz * side4 * side4, ref2, __builtin_fmaf(rz, ref1, side3 * ref0));
      side5 = __builtin_sqrtf(b2) * 1.00001f;
    } else if (tid < C::BM + C::BN) {   // h_n >= sqrt of every ratio to the reference feature (0/0 counts as 0)
      const float q = __int_as_float(side2);
      float h2 = fmaxf(q / ref0, side3 / ref1);
      if (side4 > 0.f) h2 = fmaxf(h2, side4 / ref2);
      side5 = (q > 0.f || side3 > 0.f || side4 > 0.f) ? __builtin_sqrtf(h2) * 1.00001f : 0.f;
    }
    side[5 * C::NT + tid] = side5;
  }
  gemm_epilogue<C, DENSE>(acc, ep, m0 + T, m0, n0, 0, wave, lane, smem, side, [] {});
}

// Host launcher: A = xq row-major [>= BM rows][d]; B = Wq fragment-major (op.packed = 3); optional
// outlier tiles [rows][128] row-major.
template <int BM, int NW, bool DENSE>
inline int gemm_skinny_launch_nw(const GemmOperands &op, int T, int d, int N, const GemmEpilogue &ep, hipStream_t s) {
  using C = SkinnyCfg<BM, NW>;
  const int m_tiles = (T + BM - 1) / BM;               // token tiles (A must hold m_tiles * BM rows)
  if (T <= 0 || m_tiles > 2 || N % C::BN || d % C::KC || op.packed != 3 || op.ldA != (size_t)d) return MSAE_EINVAL;
  auto kern = gemm_skinny_kernel<BM, NW, DENSE>;
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(N / C::BN, m_tiles), dim3(C::NT), C::LDS_BYTES, s, op, T, d, ep);
  return (int)hipGetLastError();
}
// The sample pass (DENSE, few tiles) runs 4-wave workgroups, two per CU: twice the workgroups and their prologue / epilogue
// phases overlap.  The main pass (THRESH) runs one 8-wave workgroup per CU: every workgroup streams the tokens' rows (xq, out
// of L2) once per 32 features of each wave, so wider workgroups halve that traffic (T = 64: 0.139 -> 0.129 ms).
// 129 ... 256 tokens (BM = 256): the main pass keeps ONE tile of 256 tokens per feature block (Wq from HBM once: 16 token groups x
// 2 feature groups of accumulators per wave, a 256-byte A chunk, B batches of one k-step); the sample pass runs the 128-token
// kernel on two token tiles.
template <int BM, bool DENSE>
inline int gemm_skinny_launch(const GemmOperands &op, int T, int d, int N, const GemmEpilogue &ep, hipStream_t s) {
  if constexpr (DENSE) return gemm_skinny_launch_nw<(BM > 128 ? 128 : BM), 4, DENSE>(op, T, d, N, ep, s);
  else return gemm_skinny_launch_nw<BM, 8, DENSE>(op, T, d, N, ep, s);
}
