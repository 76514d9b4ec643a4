// gemm_mfma96.h -- TIMING PROTOTYPE (round 5, tools/gemm96_probe.hip; not part of the product): the int8 candidate GEMM with
// 96-byte k-tiles in a THREE-slot LDS ring -- two k-tiles (96 KB) in flight instead of one (64 KB) -- against the product's
// 128-byte / 2-slot ring (csrc/gemm_mfma.h).  Same 256 x 256 tile, wave grid, MFMA, XCD-aware persistent tile walk, block issue of
// the LDS-DMA pieces, staggered issue, hand-scheduled fragment reads and THRESH epilogue; what differs:
//   * a stage is 256 x 96 B of A + 256 x 96 B of B = 48 KB; 3 slots = 144 KB; the side buffer and a SHORT candidate queue fill the rest
//     (a product version would alias them into the slot that is free during the epilogue);
//   * three k-steps per k-tile; the pieces of k-tile kt + 2 are issued in iteration kt and the wait at the top is COUNTED
//     (vmcnt(PPW): the pieces of kt + 1 may still be in flight; loads retire in order);
//   * rows are 96 B apart: six 16-B chunks, rotated by bit 3 of the row (conflict-free for the 16 consecutive rows of a b128 phase:
//     6 r mod 16 repeats with period 8, the rotation moves the second half to the other parity).
// No outlier tile, no k rotation, tile-major operands only.  Operand CONTENT is whatever the harness fills in: this measures time.
#pragma once
#include "../../multimodal-sae_amd/csrc/gemm_mfma.h"

struct Gemm96Cfg {
  static constexpr int BM = 256, BN = 256, STAGES = 3, WM = 2, WN = 4;
  static constexpr bool I8 = true, F8 = false, CERT = false, SCALED = true;
  static constexpr bool ABL_NOSTAGE = false, ABL_NOREAD = false, ABL_NOMFMA = false;
  static constexpr int NWAVES = 8, NT = 512, TM = 128, TN = 64, MI = 4, NI = 2;
  static constexpr int ROWB = 96, KS = 3;
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_RING_BYTES = STAGES * STAGE_BYTES;
  static constexpr int SIDE_SLOTS = 6, SIDE_BYTES = SIDE_SLOTS * NT * 4;
  static constexpr int QCAP = 440;
  static constexpr int LDS_BYTES = LDS_RING_BYTES + SIDE_BYTES + 16 + QCAP * 8;
  static constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVES, A_PIECES = A_BYTES / 1024;   // 48, 6, 24
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(PPW == 6 && A_PIECES % PPW == 0, "a wave's pieces are one block of one operand");
};

template <class C>
__device__ __forceinline__ void gemm96_stage_block(const unsigned char *__restrict__ src, unsigned char *dst_lds, int lane) {
  const unsigned voff = (unsigned)lane << 4;
  const unsigned d0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)dst_lds;
  const unsigned char *s1 = src + 4096;
  const unsigned d1 = d0 + 4096u;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:3072"
               :: "v"(voff), "s"(src), "s"(d0) : "memory", "m0");
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024"
               :: "v"(voff), "s"(s1), "s"(d1) : "memory", "m0");
}

template <class C>
__device__ __forceinline__ void gemm96_read_frags(i32x4 (&a)[C::MI], i32x4 (&b)[C::NI], unsigned addrA, unsigned addrB) {
  a[0] = lds_read_b128<0 * 3072>(addrA); a[1] = lds_read_b128<1 * 3072>(addrA);
  a[2] = lds_read_b128<2 * 3072>(addrA); a[3] = lds_read_b128<3 * 3072>(addrA);
  b[0] = lds_read_b128<0 * 3072>(addrB); b[1] = lds_read_b128<1 * 3072>(addrB);
}
template <class C, class F>
__device__ __forceinline__ void gemm96_compute(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA, int wr, int wc, int l31,
                                               int kh, F &&mid) {
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)sA;
  const unsigned rowA = base + (unsigned)(wr * C::TM + l31) * 96u;
  const unsigned rowB = base + (unsigned)C::A_BYTES + (unsigned)(wc * C::TN + l31) * 96u;
  const unsigned rot = (unsigned)(l31 >> 3) & 1u;     // bit 3 of the row (block offsets are multiples of 32 rows)
  unsigned off[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) off[ks] = (((unsigned)(ks * 2 + kh) + rot) % 6u) << 4;
  i32x4 a0[C::MI], b0[C::NI], a1[C::MI], b1[C::NI];
  gemm96_read_frags<C>(a0, b0, rowA + off[0], rowB + off[0]);
  gemm96_read_frags<C>(a1, b1, rowA + off[1], rowB + off[1]);
  lgkm_wait_tied<6, C>(a0, b0);
  gemm_mfma_step<C>(acc, a0, b0);
  gemm96_read_frags<C>(a0, b0, rowA + off[2], rowB + off[2]);
  mid(0);
  lgkm_wait_tied<6, C>(a1, b1);
  gemm_mfma_step<C>(acc, a1, b1);
  mid(1);
  lgkm_wait_tied<0, C>(a0, b0);
  gemm_mfma_step<C>(acc, a0, b0);
}

template <class C, int LATE_AT>
__global__ __launch_bounds__(C::NT) void gemm96_kernel(GemmOperands op, int T, int Tp, int N, int nM, int nN, GemmEpilogue ep) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  int seq = 0;                                   // flat k-tile counter; ring slot = seq % 3
  for (int tile_id = blockIdx.x; tile_id < nM * nN; tile_id += gridDim.x) {
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int l31 = lane & 31, kh = lane >> 5;
    int tm, tn;
    gemm_map_tile(tile_id, nM, nN, tm, tn);
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const bool has_next = tile_id + (int)gridDim.x < nM * nN;
    int m0n = 0, n0n = 0;
    if (has_next) {
      gemm_map_tile(tile_id + gridDim.x, nM, nN, tm, tn);
      m0n = tm * C::BM, n0n = tn * C::BN;
    }
    float side0 = 0.f, side1 = 0.f, side3 = 0.f, side4 = 0.f;
    int side2 = 1;
    float ref0 = ep.refs[0], ref1 = ep.refs[1], ref2 = ep.refs[2];
    {
      const int tid = tid_;
      if (tid < C::BM) {
        const int t = m0 + tid;
        const float v = (t < T) ? ep.tau_vals[(size_t)t * ep.tau_ld + ep.tau_col] : 0.f;
        side0 = (v > 0.f) ? v : __builtin_inff();
        if (t < T) {
          const f32x4 rc = ep.rowc[t];
          side1 = rc[0]; side3 = rc[2]; side4 = 1.f;
        }
      } else if (tid < C::BM + C::BN) {
        const int n = n0 + tid - C::BM;
        const int feat = gemm_feature(ep, n);
        side0 = ep.bias ? ep.bias[feat] : 0.f;
        const f32x4 cc = ep.colc[n];
        side1 = cc[0]; side2 = __float_as_int(cc[1]); side3 = cc[2]; side4 = cc[3];
      }
    }
    f32x16 acc[C::MI][C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
      for (int j = 0; j < C::NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int ntiles = op.nk;
    // tile-major operands, 24-KB blocks [row tile][k-tile]; waves 0-3 carry the A block, 4-7 the B block, 6 KB each
    auto stage = [&](int tm0, int tn0, int kq, int slot) {
      const bool wa = wave * C::PPW < C::A_PIECES;
      const unsigned char *o = wa ? op.A : op.B;
      const size_t rt = (size_t)(wa ? tm0 / C::BM : tn0 / C::BN);
      const size_t win = (size_t)(wave * C::PPW - (wa ? 0 : C::A_PIECES)) * 1024;
      gemm96_stage_block<C>(o + (rt * op.nk + kq) * C::A_BYTES + win, smem + slot * C::STAGE_BYTES + wave * C::PPW * 1024, lane);
    };
    // position p of THIS tile's sequence (p >= ntiles: the next tile's)
    auto stage_pos = [&](int p, int slot) -> bool {
      if (p < ntiles) { stage(m0, n0, p, slot); return true; }
      if (has_next && p - ntiles < ntiles) { stage(m0n, n0n, p - ntiles, slot); return true; }
      return false;
    };
    if (tile_id == (int)blockIdx.x) {   // later tiles: positions 0 and 1 were staged by the predecessor
      stage_pos(0, seq % 3);
      stage_pos(1, (seq + 1) % 3);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
      // the pieces of k-tile kt landed; those of kt + 1 (issued one iteration ago, six per wave) may still fly
      const bool newer = kt + 1 < ntiles || has_next;                    // wave-uniform
      if (newer) wait_vmcnt<C::PPW>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const bool late = wave >= C::NWAVES / 2;
      const int s2 = (seq + 2) % 3;
      if (!late) stage_pos(kt + 2, s2);
      const unsigned char *sA = smem + (seq % 3) * C::STAGE_BYTES;
      gemm96_compute<C>(acc, sA, wr, wc, l31, kh, [&](int pos) { if (late && pos == LATE_AT) stage_pos(kt + 2, s2); });
      ++seq;
    }
    float *side = reinterpret_cast<float *>(smem + C::LDS_RING_BYTES);
    side[tid_] = side0;
    side[C::NT + tid_] = side1;
    reinterpret_cast<int *>(side)[2 * C::NT + tid_] = side2;
    side[3 * C::NT + tid_] = side3;
    side[4 * C::NT + tid_] = side4;
    {
      float side5 = 0.f;
      if (tid_ < C::BM) {
        float b2 = side3 * ref0;
        const float rz = side1 * side1 * ep.zz12;
        b2 = __builtin_fmaf(rz * side4 * side4, ref2, __builtin_fmaf(rz, ref1, b2));
        side5 = __builtin_sqrtf(b2) * 1.00001f;
      } else if (tid_ < C::BM + C::BN) {
        const float q = __int_as_float(side2);
        float h2 = q / ref0;
        h2 = fmaxf(h2, side3 / ref1);
        if (side4 > 0.f) h2 = fmaxf(h2, side4 / ref2);
        side5 = (q > 0.f || side3 > 0.f || side4 > 0.f) ? __builtin_sqrtf(h2) * 1.00001f : 0.f;
      }
      side[5 * C::NT + tid_] = side5;
    }
    gemm_epilogue<C, false>(acc, ep, T, m0, n0, wr, wc, lane, smem, side, [] {}, 0);
  }
}

template <class C, int LATE_AT>
inline int gemm96_launch(const GemmOperands &op, int T, int Tp, int N, const GemmEpilogue &ep, hipStream_t s) {
  const int nM = Tp / C::BM, nN = N / C::BN;
  auto kern = gemm96_kernel<C, LATE_AT>;
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int n_cu = cus > 8 ? cus / 8 * 8 : 8;
  const int grid = nM * nN <= n_cu ? nM * nN : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, op, T, Tp, N, nM, nN, ep);
  return (int)hipGetLastError();
}
