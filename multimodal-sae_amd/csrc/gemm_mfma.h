// gemm_mfma.h -- candidate-pass GEMM  C[T][N] = A[T][K] * B[N][K]^T  on the gfx950 matrix cores, with
// the SAE epilogues.  Two operand types share one tile / pipeline structure:
//   bf16 : v_mfma_f32_32x32x16_bf16, f32 accumulate          (2.5 PFLOP/s dense peak)
//   int8 : v_mfma_i32_32x32x32_i8,  i32 accumulate, per-token x per-feature scales applied in the
//          epilogue; a leading "outlier" k-tile carries the few massive activation dims at a
//          coarser per-token scale (acc *= m[t] after it)   (~2x the bf16 rate, half the bytes)
// This is the dominant kernel of the fused encoder (encode_fused.hip): 2*d*N FLOP per token.  Both
// operands are K-contiguous, so A and B fragments are read the same way.
//
// Structure (template parameters BM, BN, STAGES, WM, WN; a k-tile is always 128 B per row):
//   * a workgroup of WM x WN waves computes a BM x BN tile; each wave owns (BM/WM) x (BN/WN) as
//     MI x NI blocks of 32x32 (16 accumulator VGPRs each);
//   * operand tiles travel L2 -> LDS by global_load_lds (16 B per lane, 1 KiB per wave instruction,
//     no VGPR round trip) into a ring of STAGES slots; counted s_waitcnt vmcnt + ONE raw s_barrier
//     per k-tile, so loads stay in flight across barriers;
//   * LDS image: row r of a tile at byte r*128 with its eight 16-B chunks XOR-permuted by
//     swz(r) = (r >> 1) & 7: every ds_read_b128 fragment read is bank-conflict-free
//     (SQ_LDS_BANK_CONFLICT = 0 measured).  global_load_lds writes lane-linear, so the permutation
//     is applied to the per-lane SOURCE address and again on the read;
//   * the fragment reads of a k-tile are inline-asm ds_read_b128 with COUNTED lgkmcnt waits (the
//     compiler waits lgkmcnt(0) after every group of reads): the 6 reads of k-step ks+1 are in
//     flight while the 8 MFMAs of ks execute (-5 % on the int8 pass);
//   * workgroups are persistent (one per CU) and walk their tiles as one flat k-sequence: the last
//     k-iteration of a tile already stages the first k-tile of the next one;
//   * tile -> workgroup map is XCD-aware: the 32 workgroups resident on one XCD's 32 CUs form an
//     8 (M) x 4 (N) super-tile sharing 8 A-tiles and 4 B-tiles in that XCD's private L2.
//
// Round 2 (profiles/r02_gemm_sweep_*.txt, int8 operands, 64 k-tiles per output tile): full loop 1.67 us
// per k-tile and workgroup; LDS-DMA staging alone 1.41-1.75 us (46 GB/s per CU out of L2, the same with
// plain global_load_dwordx4 -> ds_write_b128 staging: 1.72 us, so it is the L2 -> CU delivery of this
// access pattern, not the DMA instruction); MFMA + fragment reads + barriers alone 1.35-1.40 us.  The
// two-group ping-pong loop (waves 0-3 / 4-7 half a k-tile apart, one barrier per half: commit 5ad2c1e^)
// hides the DMA issue behind the partner wave's MFMAs but pays four barriers per k-tile: 1.81-1.92 us.  Issuing part of a wave's DMA pieces behind
// its own MFMA groups (2..8 pieces after k-steps 0/1/2; profiles/r02_gemm_sweep_dma_interleave.txt): -2 % to +16 %.  Register staging: 2.6-3.3 us (32 more
// live VGPRs, spills).  What paid in round 2: wave-uniform DMA base addresses (saddr form, 1 VGPR) and
// computing the epilogue's band constants after the k-loop instead of in the prologue (4.7 -> 4.27 ms).
//
// Measured on MI355X (tools/gemm_sweep, T=8192 K=4096 N=131072, random data; profiles/):
//   256x256 tile, 2-slot ring, 8 waves (2x4): bf16 1.20 PFLOP/s (1.39 on zero-filled operands: the
//   kernel is partly DVFS-bound), int8 2x that; 128x128: 1.0.  Ablations: MFMA+barriers only
//   5.4 ms, LDS-DMA staging only 5.4 ms (~12.7 TB/s L2->LDS, independent of ring depth and row
//   pitch), both 7.3 ms.  Deeper rings (half-size k-tiles x 4 slots), a two-group ping-pong
//   schedule, s_setprio, an L2 prefetch and issuing the DMA behind the first MFMA group were all
//   measured equal or worse and are not kept.
// Round 5 (profiles/r05_ab_stagger_at.txt, r05_ab_supertile.txt, r05_gemm96_prototype.txt): a wave's eight pieces are issued as ONE
// block (gemm_stage_block: two SGPR bases, the rest immediate offsets that the instruction adds to the global AND the LDS address:
// ~100 cycles instead of ~700), after which the staggered waves issue behind k-step 0 instead of 1 (-2 %); L2 warming, pre-staging
// the next tile's second k-tile, other super-tile shapes and a 96-byte / 3-slot ring (tools/tuning/gemm_mfma96.h) all measured equal.
//
// Epilogues work on the UPPER value u = v + z*sigma(t, n) of every output, v = coarse pre-activation
// (value + bias) and z*sigma the width of the error band of the operand type (encode_fused.hip):
//     z^2 sigma^2(t, n) = P_t Q_n + R_t (Si_n + M_t So_n),  R_t = sx_t^2 z^2/12, M_t = m_t^2 (int8 only)
//   DENSE  out[t][n] = u                                                      (sample pass)
//   THRESH append (feature, u) to token t's candidate list when u > tau[t].  The hot loop tests the
//          separable upper bound  v + h_n B_t > tau[t]  (one fma more than a plain compare):
//            B_t = z sigma of token t against a REFERENCE feature (refs = typical Q, Si, So),
//            h_n^2 = max(Q_n/Qr, Si_n/Sir, So_n/Sor)  =>  h_n B_t >= z sigma(t, n) for every pair;
//          survivors are queued in LDS and the exact u of each is computed when the queue is flushed
#pragma once
#include <cstdlib>

#include "common.h"
#include "tuning.h"


struct GemmEpilogue {
  const float *bias;             // b_enc
  int bias_stride, bias_off;     // feature of column n is n*bias_stride + bias_off ...
  int skip_stride, skip_off;     // ... or, with skip_stride = S > 0, the n-th feature f with f % S != skip_off (gemm_feature)
  float *dense; int ld_dense;    // DENSE
  const float *tau_vals; int tau_ld, tau_col;   // THRESH: tau[t] = tau_vals[t*tau_ld + tau_col]
  int *cnt; unsigned long long *cand; int cap;  // candidate lists
  // A token's list can be split into `segs` (0 / 1: one) equal segments with their own counters cnt[t*segs + s]: workgroup b
  // appends to segment b % segs.  A batch of few tokens has hundreds of appends per counter (~650 per token, all of the launch's
  // workgroups on <= 256 addresses); same-address atomics serialise in L2, ~45 ns each: 0.03 ms of a 0.13 ms pass.
  int segs;
  int skip_a, skip_b;            // features never emitted (hook edits replace their latents)
  // error-band constants (encode_fused.hip): per token (sx, m as float, P, -), per COLUMN of this
  // launch (sw, Q, Si, So).  int8: value = float(acc) * sx[t] * sw[n]; bf16 ignores sx, sw, Si, So.
  const f32x4 *rowc, *colc;
  const float *refs;             // (Qr, Sir, Sor): the reference feature of the separable bound
  float zz12;                    // z^2 / 12
  // Subtractive dither (round 6; encode_defs.h): row_e[t] = (E_t, m_t) -- the token's integer correction of the weights' shared
  // dither and its outlier multiplier: acc = acc_outlier_tile * m - E in the multiply that scales the outlier tile anyway -- and
  // col_ds[n] = sw_n D_n, the feature's correction of the activations' shared dither: v = (acc sw - Ds) sx + b, the same two
  // packed instructions as (acc (sx sw) + b).  Both null: no correction (m from rowc[t][1]).  int8 only.
  const int2 *row_e;
  const float *col_ds;
  unsigned long long *timeline;  // tuning builds (tuning.h, MSAE_TL): s_memtime stamps of workgroup 0 / wave 0; null in the product
};

// Operands of one launch.  A rows are tokens, B rows are features; ld* in BYTES.
struct GemmOperands {
  const unsigned char *A, *B;
  size_t ldA, ldB;
  int nk;                        // k-tiles of 128 B per row
  // int8 only: optional leading outlier tile (one k-tile, rows 128 B apart); the accumulators are
  // multiplied by the token's integer multiplier m (GemmEpilogue::rowc[t][1]) after it
  const unsigned char *Ao, *Bo;
  const int *n_out;              // device: number of outlier dims actually present (the tile is compact from column 0)
  // Tile-major operands (round 3): A and B are stored [row tile of BM / BN rows][k-tile][row in tile][128 B] with the LDS
  // image's chunk permutation already applied (position p of row r holds the row's 16-B chunk p ^ swz(r)), so a 1-KiB
  // staging piece is ONE contiguous kilobyte -- eight consecutive 128-B lines instead of eight lines a leading
  // dimension apart -- and every lane's source offset is lane * 16.  Measured on the stream alone (tools/dma_depth,
  // profiles/r03_dma_tile_major.txt): 43.2 -> 47.7 GB/s per CU with the 2 x 128 B ring, 50.6 -> 65.9 with 4 x 64 B.
  int packed;
  // Certified pass (encode_cert.h): > 0 = k-tiles per PLANE (d / 128).  A and B are tile-major with 2 * cert k-tiles per row tile
  // (plane 0 = lo, plane 1 = hi); nk = 3 * cert; k-tile q of the walk multiplies A plane (q < cert ? lo : hi) with B plane
  // (cert <= q < 2 cert ? lo : hi), and the accumulators are divided by 128 (rounded) in front of k-tile 2 * cert:
  // acc = x_hi . W_hi + round((x_lo . W_hi + x_hi . W_lo) / 128).  The token's rowc[1] is the band's M_t (no outlier tile).
  int cert;
};

// FLAGS:
//   (tuning only; results invalid when an ABL bit is set)
//   bit 2 ABL_NOSTAGE  skip the LDS-DMA staging in the loop
//   bit 3 ABL_NOREAD   read the fragments once and reuse them
//   bit 4 ABL_NOMFMA   issue no MFMA
//   bit 5 CERT         the certified three-segment pass (encode_cert.h); I8 only
//   bit 6 F8           fp8 (e4m3) operands, f32 accumulate (I8 = false)
template <int BM_, int BN_, int STAGES_, int WM_, int WN_, bool I8_ = false, int FLAGS_ = 0>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, STAGES = STAGES_, WM = WM_, WN = WN_;
  static constexpr bool I8 = I8_;
  static constexpr bool ABL_NOSTAGE = FLAGS_ & 4, ABL_NOREAD = FLAGS_ & 8, ABL_NOMFMA = FLAGS_ & 16;
  static constexpr bool CERT = FLAGS_ & 32;       // certified pass (GemmOperands::cert): its own instantiation, the product kernel is untouched
  // fp8 candidate pass (BASELINE configs[4]: "fp8 MFMA encoder path"): one byte per operand element exactly like int8 -- the SAME
  // tile-major operands, LDS image, ring and hand-scheduled fragment reads -- multiplied by v_mfma_f32_32x32x16_fp8_fp8 (two per
  // 16-byte fragment pair: the low and the high 8 bytes of the lanes) into f32 accumulators; per-token / per-feature scales in
  // the epilogue as on the int8 path, the error band of the bf16 path's form (relative roundings).  I8_ must be false.
  static constexpr bool F8 = FLAGS_ & 64;
  static_assert(!(I8_ && (FLAGS_ & 64)), "fp8 accumulates in f32");
  static constexpr bool SCALED = I8_ || F8;       // value = acc * sx[t] * sw[n] + bias (else: acc + bias)
  static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
  static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  static constexpr int ROWB = 128;               // bytes per tile row: 64 bf16 or 128 int8
  static constexpr int KS = 4;                   // MFMA k-steps per tile (32 B per lane-half each)
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_RING_BYTES = STAGES * STAGE_BYTES;
  // behind the ring: side buffer of epilogue constants, then the THRESH epilogue's candidate queue.
  // Neither overlaps the ring: the next tile's first k-tile is already landing in it meanwhile.
  // side buffer, one float per thread and slot: tau|bias, sx|sw, m (int, parked early)|Q, P|Si, m|So, B|h
  // (row threads | column threads)
  // slot 6 (round 6): E (int) | Ds
  static constexpr int SIDE_SLOTS = 7;
  static constexpr int SIDE_BYTES = SIDE_SLOTS * NT * 4;
  static constexpr int QCAP = 2296;              // (2304 before slot 6: the budget below is the CU's whole LDS)
  static constexpr int LDS_BYTES = LDS_RING_BYTES + SIDE_BYTES + 16 + QCAP * 8;
  static_assert(STAGES == 2, "the flat cross-tile k-sequence below is written for a 2-slot ring");
  static_assert(BM + BN <= NT, "one thread per tile row and column fetches the epilogue constants");
  static constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVES;  // 1-KiB pieces per wave
  static constexpr int A_PIECES = A_BYTES / 1024;
  static_assert(PIECES % NWAVES == 0, "stage must split evenly over the waves");
  static_assert(TM % 32 == 0 && TN % 32 == 0, "tile shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ int gemm_swz(int row) { return (row >> 1) & 7; }

// LDS-DMA staging.  A piece = 8 tile rows x 128 B = one wave instruction (16 B per lane).  Its global
// address is a WAVE-UNIFORM base (operand + first row of the piece + k offset: SGPRs) plus a 32-bit
// per-lane offset (row within the piece x leading dimension + swizzled 16-B chunk) that depends only
// on the parity of the piece and on the leading dimension -- four VGPRs for the whole kernel instead
// of one 64-bit address per piece.  (Tiles are always full: Tp % BM == 0 and N % BN == 0.)
struct GemmStageLane {
  unsigned off;            // byte offset of this lane inside an EVEN piece (A and B share the leading dimension);
};                         // an odd piece flips chunk bit 2: swz(row + 8) = swz(row) ^ 4, i.e. off ^ 64
__device__ __forceinline__ GemmStageLane gemm_stage_lane(int lane, unsigned ld) {
  const unsigned r_in = lane >> 3, c_in = lane & 7;
  GemmStageLane l;
  l.off = r_in * ld + ((c_in ^ ((r_in >> 1) & 7)) << 4);   // ld % 128 == 0: bit 6 belongs to the chunk index
  return l;
}
// pieces [first, first + COUNT) of one k-tile (byte offset kbyte in each row) into ring slot `slot`.
// Branch-free on purpose: control flow between two LDS-DMA instructions makes the compiler's wait-count
// pass drain vmcnt before each of them.
template <class C, int COUNT>
__device__ __forceinline__ void gemm_stage_pieces(const unsigned char *__restrict__ A,
                                                  const unsigned char *__restrict__ B, size_t ld, int m0,
                                                  int n0, size_t kbyte, unsigned char *lds, int slot, int first,
                                                  const GemmStageLane &sl) {
  unsigned char *base = lds + slot * C::STAGE_BYTES;
#pragma unroll
  for (int i = 0; i < COUNT; ++i) {
    const int piece = first + i;                      // wave-uniform
    const bool isA = piece < C::A_PIECES;
    const int pl = isA ? piece : piece - C::A_PIECES;  // piece index inside its operand tile
    const unsigned char *op0 = isA ? A : B;
    const int r0 = (isA ? m0 : n0) + pl * 8;
    const unsigned voff = sl.off ^ ((unsigned)(pl & 1) << 6);
    const unsigned char *sbase = op0 + (size_t)r0 * ld + kbyte;         // wave-uniform: SGPR pair
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(base + piece * 1024);
    // saddr + 32-bit voffset form, LDS base in M0.  Written as asm because the builtin materialises a
    // 64-bit VGPR address per piece (hoisted out of the k-loop: ~2 VGPRs per piece).  No builtin LDS-DMA
    // is left in this kernel, so the compiler never holds a value of its own in M0.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
  }
}
// tile-major operands: tileA / tileB = the 32-KB block of this k-tile of the A / B row tile (wave-uniform)
template <class C>
__device__ __forceinline__ void gemm_stage_packed(const unsigned char *__restrict__ tileA,
                                                  const unsigned char *__restrict__ tileB, unsigned char *lds,
                                                  int slot, int wave, int lane) {
  unsigned char *base = lds + slot * C::STAGE_BYTES;
  const unsigned voff = (unsigned)lane << 4;
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    const int piece = wave * C::PPW + i;               // wave-uniform
    const bool isA = piece < C::A_PIECES;
    const unsigned char *sbase = isA ? tileA + piece * 1024 : tileB + (piece - C::A_PIECES) * 1024;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(base + piece * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
  }
}

// Tile-major operands, the wave's share as ONE block: with A_PIECES % PPW == 0 a wave's PPW pieces are PPW contiguous kilobytes of
// ONE operand's 32-KB k-tile block and land in PPW contiguous kilobytes of the ring slot, and the instruction's immediate offset
// is added to BOTH the global and the LDS address -- so four pieces share one SGPR base and one M0 (offsets 0 / 1 / 2 / 3 KiB).
// Scalar work per wave and k-tile: ~12 SALU instead of ~90 (per piece: two 64-bit candidate bases, the A / B select, the LDS
// address, M0).  `src` = the wave's first source byte, `dst` = its first LDS byte (both wave-uniform).
template <class C>
struct GemmWaveBlock {
  static constexpr bool OK = C::A_PIECES % C::PPW == 0 && C::PPW % 4 == 0;
};
template <class C>
__device__ __forceinline__ void gemm_stage_block(const unsigned char *__restrict__ src, unsigned char *dst_lds, int lane) {
  const unsigned voff = (unsigned)lane << 4;
#pragma unroll
  for (int g = 0; g < C::PPW / 4; ++g) {
    const unsigned char *sbase = src + g * 4096;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(dst_lds + g * 4096);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072"
                 :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
  }
}

// this wave's even share (PPW pieces) of one k-tile
template <class C>
__device__ __forceinline__ void gemm_stage(const unsigned char *__restrict__ A,
                                           const unsigned char *__restrict__ B, size_t ld, int m0,
                                           int n0, size_t kbyte, unsigned char *lds, int slot, int wave,
                                           const GemmStageLane &sl) {
  gemm_stage_pieces<C, C::PPW>(A, B, ld, m0, n0, kbyte, lds, slot, wave * C::PPW, sl);
}

__device__ __forceinline__ i32x4 gemm_frag(const unsigned char *tile, int row, int chunk) {
  return *reinterpret_cast<const i32x4 *>(tile + row * 128 + ((chunk ^ gemm_swz(row)) << 4));
}

// tile id -> (tm, tn); see header.  Falls back to M-fastest order when the grid does not factor.
__device__ __forceinline__ void gemm_map_tile(int b, int nM, int nN, int &tm, int &tn) {
  constexpr int GM = MSAE_GEMM_GM, GN = MSAE_GEMM_GN;   // (tuning.h: 8 x 4)
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

// one k-tile of MFMAs out of the LDS images sA / sB
template <class C>
__device__ __forceinline__ void gemm_mfma_step(f32x16 (&acc)[C::MI][C::NI], const i32x4 (&a)[C::MI],
                                               const i32x4 (&b)[C::NI]) {
  // (issue order: i outer -- consecutive MFMAs share the A fragment; j outer, four in a row on one B fragment, measured the same:
  // 3.84-3.86 vs 3.82-3.87 ms, NOTEBOOK.md)
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      if constexpr (C::I8) {  // the accumulator registers hold i32 on this path
        acc[i][j] = __builtin_bit_cast(
            f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], __builtin_bit_cast(i32x16, acc[i][j]), 0, 0, 0));
      } else if constexpr (C::F8) {
        typedef long i64x2 __attribute__((ext_vector_type(2)));
        const i64x2 a2 = __builtin_bit_cast(i64x2, a[i]), b2 = __builtin_bit_cast(i64x2, b[j]);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[0], b2[0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[1], b2[1], acc[i][j], 0, 0, 0);
      } else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                            __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
}

// ---- hand-scheduled k-tile ----------------------------------------------------------------------------
// The compiler waits lgkmcnt(0) after every group of fragment reads, so a read issued for the NEXT
// k-step stalls the MFMAs of the current one.  Here the ds_read_b128 are inline asm (invisible to
// the compiler's wait-count pass) with counted waits: the 6 reads of k-step ks+1 stay in flight
// while the 8 MFMAs of ks execute.  The "+v" ties make every MFMA depend on the wait that covers
// its operands.
template <int IMM>
__device__ __forceinline__ i32x4 lds_read_b128(unsigned addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
template <int N, class C>
__device__ __forceinline__ void lgkm_wait_tied(i32x4 (&a)[C::MI], i32x4 (&b)[C::NI]) {
  static_assert(C::MI == 4 && C::NI == 2, "operand list below is written for a 128x64 wave tile");
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])
               : "n"(N)
               : "memory");
}
template <class C>
__device__ __forceinline__ void gemm_read_frags(i32x4 (&a)[C::MI], i32x4 (&b)[C::NI], unsigned addrA,
                                                unsigned addrB) {
  a[0] = lds_read_b128<0 * 4096>(addrA); a[1] = lds_read_b128<1 * 4096>(addrA);
  a[2] = lds_read_b128<2 * 4096>(addrA); a[3] = lds_read_b128<3 * 4096>(addrA);
  b[0] = lds_read_b128<0 * 4096>(addrB); b[1] = lds_read_b128<1 * 4096>(addrB);
}
// `mid(pos)` is called at every issue position of the k-tile: -1 behind the first fragment reads, 0 / 1 / 2 behind the MFMAs of
// that k-step (the next reads in flight across it).  The waves that issue the next k-tile's LDS-DMA inside the compute phase
// instead of before their first MFMA do it at position MSAE_GEMM_STAGGER_AT (gemm_kernel: stagger)
template <class C, class F>
__device__ __forceinline__ void gemm_compute_asm(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA,
                                                 int wr, int wc, int l31, int kh, F &&mid) {
  static_assert(C::KS == 4, "four k-steps per tile");
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)sA;
  const unsigned rowA = base + (unsigned)(wr * C::TM + l31) * 128u;
  const unsigned rowB = base + (unsigned)C::A_BYTES + (unsigned)(wc * C::TN + l31) * 128u;
  const unsigned sw = (unsigned)gemm_swz(l31);       // same swizzle for every 32-row block of A and B
  unsigned off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off[ks] = (((unsigned)(ks * 2 + kh)) ^ sw) << 4;
  i32x4 a0[C::MI], b0[C::NI], a1[C::MI], b1[C::NI];
  gemm_read_frags<C>(a0, b0, rowA + off[0], rowB + off[0]);
  gemm_read_frags<C>(a1, b1, rowA + off[1], rowB + off[1]);
  mid(-1);                         // (behind the first reads, in front of the first MFMA: measured worse than any other position)
  lgkm_wait_tied<6, C>(a0, b0);
  gemm_mfma_step<C>(acc, a0, b0);
  gemm_read_frags<C>(a0, b0, rowA + off[2], rowB + off[2]);
  mid(0);
  lgkm_wait_tied<6, C>(a1, b1);
  gemm_mfma_step<C>(acc, a1, b1);
  gemm_read_frags<C>(a1, b1, rowA + off[3], rowB + off[3]);
  mid(1);
  lgkm_wait_tied<6, C>(a0, b0);
  gemm_mfma_step<C>(acc, a0, b0);
  mid(2);
  lgkm_wait_tied<0, C>(a1, b1);
  gemm_mfma_step<C>(acc, a1, b1);
}
template <class C>
__device__ __forceinline__ void gemm_compute_asm(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA,
                                                 int wr, int wc, int l31, int kh) {
  gemm_compute_asm<C>(acc, sA, wr, wc, l31, kh, [](int) {});
}

// Compact outlier tile (at most 32 outlier dims: one k-step).  The k-loop is bound by the L2 -> LDS delivery, and a
// full 64 KB tile of which 32 B per row carry data is 3 % of an output tile's bytes for nothing: only bytes [0, 32) of
// each 128-B row are staged -- 16 pieces of 32 rows x 32 B, two per wave -- into a dense [512][32 B] image (A rows,
// then B rows) at the head of the slot.
template <class C>
__device__ __forceinline__ void gemm_stage_lead_compact(const unsigned char *__restrict__ Ao,
                                                        const unsigned char *__restrict__ Bo, int m0, int n0,
                                                        unsigned char *lds, int slot, int wave, int lane) {
  static_assert(C::BM == 256 && C::BN == 256 && C::NWAVES == 8, "16 pieces over 8 waves");
  unsigned char *base = lds + slot * C::STAGE_BYTES;
  const unsigned voff = (unsigned)(lane >> 1) * 128u + (unsigned)(lane & 1) * 16u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;                    // 0..7: A rows 32 piece .., 8..15: B rows
    const bool isA = piece < 8;
    const int pl = isA ? piece : piece - 8;
    const unsigned char *sbase = (isA ? Ao + (size_t)m0 * 128 : Bo + (size_t)n0 * 128) + (size_t)pl * 32 * 128;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(base + piece * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
  }
}
template <class C>
__device__ __forceinline__ void gemm_compute_lead_compact(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA,
                                                          int wr, int wc, int l31, int kh) {
  const unsigned char *sB = sA + C::BM * 32;
  i32x4 a[C::MI], b[C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i) a[i] = *reinterpret_cast<const i32x4 *>(sA + (wr * C::TM + i * 32 + l31) * 32 + kh * 16);
#pragma unroll
  for (int j = 0; j < C::NI; ++j) b[j] = *reinterpret_cast<const i32x4 *>(sB + (wc * C::TN + j * 32 + l31) * 32 + kh * 16);
  gemm_mfma_step<C>(acc, a, b);
}

// the outlier k-tile: only its first `nks` k-steps hold data (32 outlier dims per k-step), the rest is zero
template <class C>
__device__ __forceinline__ void gemm_compute_lead(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA,
                                                  const unsigned char *sB, int wr, int wc, int l31, int kh, int nks) {
  for (int ks = 0; ks < nks; ++ks) {
    const int chunk = ks * 2 + kh;
    i32x4 a[C::MI], b[C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) a[i] = gemm_frag(sA, wr * C::TM + i * 32 + l31, chunk);
#pragma unroll
    for (int j = 0; j < C::NI; ++j) b[j] = gemm_frag(sB, wc * C::TN + j * 32 + l31, chunk);
    gemm_mfma_step<C>(acc, a, b);
  }
}

template <class C>
__device__ __forceinline__ void gemm_compute(f32x16 (&acc)[C::MI][C::NI], const unsigned char *sA,
                                             const unsigned char *sB, int wr, int wc, int l31, int kh,
                                             i32x4 (&abl_a)[C::MI], i32x4 (&abl_b)[C::NI]) {
  if constexpr (!C::ABL_NOREAD && !C::ABL_NOMFMA) {
    gemm_compute_asm<C>(acc, sA, wr, wc, l31, kh);
    return;
  }
  // ablation builds only (tools/gemm_sweep): compiler-scheduled reads / no reads / no MFMA
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int chunk = ks * 2 + kh;
    i32x4 a[C::MI], b[C::NI];
    if constexpr (C::ABL_NOREAD) {
#pragma unroll
      for (int i = 0; i < C::MI; ++i) { a[i] = abl_a[i]; asm volatile("" : "+v"(a[i])); }
#pragma unroll
      for (int j = 0; j < C::NI; ++j) { b[j] = abl_b[j]; asm volatile("" : "+v"(b[j])); }
    } else {
#pragma unroll
      for (int i = 0; i < C::MI; ++i) a[i] = gemm_frag(sA, wr * C::TM + i * 32 + l31, chunk);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) b[j] = gemm_frag(sB, wc * C::TN + j * 32 + l31, chunk);
    }
    if constexpr (C::ABL_NOMFMA) {
#pragma unroll
      for (int i = 0; i < C::MI; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int j = 0; j < C::NI; ++j) asm volatile("" ::"v"(b[j]));
    } else {
      gemm_mfma_step<C>(acc, a, b);
    }
  }
}

// Epilogue.  DENSE stores every value.  THRESH compares with the per-token threshold; survivors are
// rare (~1 per 200 outputs pass the hot loop's separable bound) but each needs a slot in its token's
// candidate list, i.e. a RETURNING global atomic (~2 us round trip).  Doing that inline serialises ~30
// round trips per wave -- as long as the whole k-loop.  So survivors are first queued in LDS (its own
// region behind the ring) and then flushed, one queue entry per lane: the global atomics of the whole
// workgroup are in flight together.  The hot loop itself touches no LDS queue: hit masks per 32x32
// block first (4.5 VALU per output), then ONE queue reservation per lane and tile, then the pushes
// (round 3; 15.4k -> 7.8k cycles per tile on the tile timeline, profiles/r03_epilogue_batch.txt --
// worth 0-1 % of wall time only, because the kernel runs at the package power limit: the clock drops
// as the cycle count does, profiles/r03_power.txt).
// z^2 sigma^2 of pair (row, col) from the side buffer (same expression as band_sq in encode_fused.hip)
template <class C>
__device__ __forceinline__ float gemm_band_sq(const float *side, int row, int col, float zz12) {
  const float *row_c = side, *col_c = side + C::BM;
  const float pz = row_c[3 * C::NT + row];
  if constexpr (!C::I8 && !C::F8) return pz * col_c[2 * C::NT + col];
  const float rs = row_c[C::NT + row], mf = row_c[4 * C::NT + row];
  const float rz = rs * rs * zz12;
  return __builtin_fmaf(pz, col_c[2 * C::NT + col], __builtin_fmaf(rz * mf * mf, col_c[4 * C::NT + col], rz * col_c[3 * C::NT + col]));
}

// append one candidate key to token t's list (segment of this workgroup)
__device__ __forceinline__ void gemm_push_candidate(const GemmEpilogue &ep, int t, unsigned long long key) {
  if (ep.segs > 1) {
    const int sg = (int)(blockIdx.x % (unsigned)ep.segs), scap = ep.cap / ep.segs;
    const int gslot = atomicAdd(ep.cnt + (size_t)t * ep.segs + sg, 1);
    if (gslot < scap) ep.cand[(size_t)t * ep.cap + (size_t)sg * scap + gslot] = key;
  } else {
    const int gslot = atomicAdd(ep.cnt + t, 1);
    if (gslot < ep.cap) ep.cand[(size_t)t * ep.cap + gslot] = key;
  }
}

// feature id of column n of this launch
__device__ __forceinline__ int gemm_feature(const GemmEpilogue &ep, int n) {
  if (ep.skip_stride) {
    const int g = n / (ep.skip_stride - 1), p = n - g * (ep.skip_stride - 1);
    return g * ep.skip_stride + p + (p >= ep.skip_off ? 1 : 0);
  }
  return n * ep.bias_stride + ep.bias_off;
}

template <class C, bool DENSE, class F>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[C::MI][C::NI], const GemmEpilogue &ep, int T,
                                              int m0, int n0, int wr, int wc, int lane,
                                              unsigned char *smem, const float *side, F &&after_barrier, int tl_tile = 0) {
  const int l31 = lane & 31, kh = lane >> 5;
  constexpr int QCAP = C::QCAP;
  unsigned *q_count = reinterpret_cast<unsigned *>(smem + C::LDS_RING_BYTES + C::SIDE_BYTES);
  unsigned long long *queue = reinterpret_cast<unsigned long long *>(smem + C::LDS_RING_BYTES + C::SIDE_BYTES + 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // side[] written; previous tile's flush done
  after_barrier();
  if constexpr (!DENSE) {
    if (threadIdx.x == 0) *q_count = 0u;
    __syncthreads();
  }
  MSAE_TL(7);
  // side[s*NT + tid]: slot s of row tid (tid < BM) or of column tid - BM
  const float *row_c = side, *col_c = side + C::BM;
  // column constants of this lane's NI columns stay in registers across the row loops
  float c_bias[C::NI], c_sw[C::NI], c_h[C::NI];
  [[maybe_unused]] float c_ds[C::NI];                  // -sw_n D_n (subtractive dither; 0 otherwise)
  bool c_live[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const int col = wc * C::TN + j * 32 + l31;         // column inside the tile
    c_bias[j] = col_c[col];
    c_sw[j] = col_c[C::NT + col];
    if constexpr (C::I8 && !C::CERT) c_ds[j] = -col_c[6 * C::NT + col];
    c_h[j] = col_c[5 * C::NT + col];
    const int feat = gemm_feature(ep, n0 + col);
    c_live[j] = (feat != ep.skip_a) && (feat != ep.skip_b);
  }
  // C[i][n] of a 32x32 block: n = lane&31, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (msae_tuning::ABL_NOEPI && !DENSE && ep.cap != -12345) { asm volatile("" ::"v"(acc[0][0][0])); } else   // (tuning builds: skip the element loop)
  // THRESH.  One pass over the wave's MI x NI blocks without any LDS round trip: the value v replaces the accumulator in its register,
  // the sign of (v + h_n B_t) - tau goes through a 1-instruction shift register (v_alignbit) into a 16-bit hit mask per block.
  // Then ONE queue reservation per lane for all its hits of this tile, then the pushes (v picked out of the 16 registers
  // by a select tree on the hit's position).
  if constexpr (!DENSE) {
    unsigned hits[C::MI][C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
      float tau[16], rs[16], bt[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        tau[e] = row_c[row];
        rs[e] = C::SCALED ? row_c[C::NT + row] : 0.f;
        bt[e] = row_c[5 * C::NT + row];
      }
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {                                      // two outputs per packed-f32 instruction
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 tau2 = {tau[e], tau[e + 1]}, rs2 = {rs[e], rs[e + 1]}, bt2 = {bt[e], bt[e + 1]};
          f32x2 v;
          if constexpr (C::I8 && !C::CERT) {
            // v = (acc sw_n - sw_n D_n) sx_t + b_n: two packed fmas, as many instructions as acc (sx sw) + b
            const f32x2 a = {(float)__builtin_bit_cast(i32x16, acc[i][j])[e], (float)__builtin_bit_cast(i32x16, acc[i][j])[e + 1]};
            const f32x2 sw2 = {c_sw[j], c_sw[j]}, ds2 = {c_ds[j], c_ds[j]}, b2 = {c_bias[j], c_bias[j]};
            v = __builtin_elementwise_fma(__builtin_elementwise_fma(a, sw2, ds2), rs2, b2);
          } else if constexpr (C::I8) {
            const f32x2 a = {(float)__builtin_bit_cast(i32x16, acc[i][j])[e], (float)__builtin_bit_cast(i32x16, acc[i][j])[e + 1]};
            v = a * (rs2 * c_sw[j]) + c_bias[j];
          } else if constexpr (C::F8) {
            v = f32x2{acc[i][j][e], acc[i][j][e + 1]} * (rs2 * c_sw[j]) + c_bias[j];
          } else {
            v = f32x2{acc[i][j][e], acc[i][j][e + 1]} + c_bias[j];
          }
          acc[i][j][e] = v.x;
          acc[i][j][e + 1] = v.y;
          const f32x2 ch2 = {c_h[j], c_h[j]};
          const f32x2 dlt = __builtin_elementwise_fma(ch2, bt2, v) - tau2;      // >= +0 iff the bound reaches tau (NaN: either)
          m = __builtin_amdgcn_alignbit(m, __float_as_uint(dlt.x), 31);        // m = m << 1 | sign(dlt)
          m = __builtin_amdgcn_alignbit(m, __float_as_uint(dlt.y), 31);
        }
        hits[i][j] = c_live[j] ? (~m & 0xFFFFu) : 0u;                          // bit 15 - e: output e is a hit
      }
    }
    unsigned total = 0;
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
      for (int j = 0; j < C::NI; ++j) total += (unsigned)__builtin_popcount(hits[i][j]);
    if (total) {
      unsigned slot;
      {  // the hardware serialises the lanes of one LDS atomic on one address (asm: the compiler would turn atomicAdd into
         // a scalar loop over the active lanes)
        const unsigned qa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)q_count;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(qa), "v"(total) : "memory");
      }
#pragma unroll
      for (int i = 0; i < C::MI; ++i) {
#pragma unroll
        for (int j = 0; j < C::NI; ++j) {
          unsigned h = hits[i][j];
          const int col = wc * C::TN + j * 32 + l31;
          while (h) {
            const int p = 31 - __builtin_clz(h);                                // highest set bit first = lowest e first
            h &= ~(1u << p);
            const int e = 15 - p;
            // v = acc[i][j][e] by a bit-select tree (a ?: tree is turned into a scratch array indexed by e)
            const unsigned b0 = 0u - (e & 1), b1 = 0u - ((e >> 1) & 1), b2 = 0u - ((e >> 2) & 1), b3 = 0u - ((e >> 3) & 1);
            auto sel = [](unsigned mk, unsigned hi, unsigned lo) { return (hi & mk) | (lo & ~mk); };
            unsigned s8[8], s4[4], s2[2];
#pragma unroll
            for (int q = 0; q < 8; ++q) s8[q] = sel(b0, __float_as_uint(acc[i][j][2 * q + 1]), __float_as_uint(acc[i][j][2 * q]));
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = sel(b1, s8[2 * q + 1], s8[2 * q]);
#pragma unroll
            for (int q = 0; q < 2; ++q) s2[q] = sel(b2, s4[2 * q + 1], s4[2 * q]);
            const float v = __uint_as_float(sel(b3, s2[1], s2[0]));
            const int row = wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (slot < QCAP) {
              queue[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(row << 16 | col);
            } else {                                                            // queue full: slow path
              const float u = v + __builtin_sqrtf(gemm_band_sq<C>(side, row, col, ep.zz12));
              if (u > row_c[row]) {
                const int t = m0 + row;
                const int feat = gemm_feature(ep, n0 + col);
                gemm_push_candidate(ep, t, ((unsigned long long)f32_order_key(u) << 32) | (unsigned)(0x7FFFFFFF - feat));
              }
            }
            ++slot;
          }
        }
      }
    }
  } else {
    // DENSE: every value + its band, C[i][n] of a 32x32 block: n = lane&31, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        const int col = wc * C::TN + j * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh, t = m0 + row;
          float v;
          if constexpr (C::I8 && !C::CERT)
            v = __builtin_fmaf(__builtin_fmaf((float)__builtin_bit_cast(i32x16, acc[i][j])[e], c_sw[j], c_ds[j]), row_c[C::NT + row], c_bias[j]);
          else if constexpr (C::I8) v = (float)__builtin_bit_cast(i32x16, acc[i][j])[e] * (row_c[C::NT + row] * c_sw[j]) + c_bias[j];
          else if constexpr (C::F8) v = acc[i][j][e] * (row_c[C::NT + row] * c_sw[j]) + c_bias[j];
          else v = acc[i][j][e] + c_bias[j];
          if (t < T) ep.dense[(size_t)t * ep.ld_dense + n0 + col] = v + __builtin_sqrtf(gemm_band_sq<C>(side, row, col, ep.zz12));
        }
      }
    }
  }
  MSAE_TL(4);
  if constexpr (!DENSE) {
    __syncthreads();
    const unsigned nq = msae_tuning::ABL_NOFLUSH ? 0u : (*q_count < QCAP ? *q_count : QCAP);
    for (unsigned q = threadIdx.x; q < nq; q += C::NT) {
      const unsigned long long e = queue[q];
      const int row = (int)((e >> 16) & 0xFFFFu), col = (int)(e & 0xFFFFu);
      const float v = __uint_as_float((unsigned)(e >> 32));
      const float u = v + __builtin_sqrtf(gemm_band_sq<C>(side, row, col, ep.zz12));   // the exact upper value
      if (!(u > row_c[row])) continue;
      const int t = m0 + row;
      const int feat = gemm_feature(ep, n0 + col);
      gemm_push_candidate(ep, t, ((unsigned long long)f32_order_key(u) << 32) | (unsigned)(0x7FFFFFFF - feat));
    }
  }
  MSAE_TL(5);
}

template <class C, bool DENSE>
__global__ __launch_bounds__(C::NT) void gemm_kernel(GemmOperands op, int T, int Tp, int N, int nM, int nN,
                                                     GemmEpilogue ep) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // persistent workgroups: the grid is one workgroup per CU (a multiple of 8, so a workgroup's
  // tiles keep its XCD's super-tile affinity) and each walks tile ids blockIdx.x, +gridDim.x, ...
  // The k-tiles of all its output tiles form ONE flat sequence through the 2-slot ring: the last
  // iteration of a tile already stages the first k-tile of the next one, whose latency is then
  // hidden behind the epilogue.
  int seq = 0;                                   // flat k-tile counter; ring slot = seq & 1
  [[maybe_unused]] int tl_tile = -1;
  for (int tile_id = blockIdx.x; tile_id < nM * nN; tile_id += gridDim.x) {
  ++tl_tile;
  MSAE_TL(0);
  // the thread id is made opaque per tile: otherwise every lane-dependent address of the
  // prologue and the epilogue is hoisted out of this loop and lives across the k-loop (spills)
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

  // Row / column constants of the epilogue: fetched NOW into registers, parked in the LDS side buffer
  // after the k-loop, so the epilogue never waits on global memory.  (Issuing these loads behind the
  // tile's first barrier instead -- so that the first vmcnt(0) does not wait for them -- measured 5 %
  // SLOWER: their address arithmetic then sits between the DMA issue and the first MFMAs of the tile.)
  //   threads [0, BM)      : tau (THRESH) and (sx, m, P) of row m0 + tid
  //   threads [BM, BM+BN)  : bias and (sw, Q, Si, So) of column n0 + tid - BM
  float side0 = 0.f, side1 = 0.f, side3 = 0.f, side4 = 0.f;
  int side2 = 1;
  [[maybe_unused]] int side6 = 0;                // E_t (rows, int) | Ds_n (columns, float bits)
  float ref0 = 1.f, ref1 = 1.f, ref2 = 1.f;
  {
    if constexpr (!DENSE) { ref0 = ep.refs[0]; ref1 = ep.refs[1]; ref2 = ep.refs[2]; }   // consumed after the k-loop
    const int tid = tid_;
    if (tid < C::BM) {
      const int t = m0 + tid;
      if constexpr (!DENSE) {
        const float v = (t < T) ? ep.tau_vals[(size_t)t * ep.tau_ld + ep.tau_col] : 0.f;
        side0 = (v > 0.f) ? v : __builtin_inff();     // degenerate / padded token: emit nothing
      }
      if (t < T) {
        const f32x4 rc = ep.rowc[t];
        side1 = rc[0];
        side3 = rc[2];
        side4 = 1.f;
        if (C::I8 && op.Ao != nullptr) { side2 = (int)rc[1]; side4 = rc[1]; }
        if constexpr (C::CERT || C::F8) side4 = rc[1];
        if constexpr (C::I8 && !C::CERT) {
          if (ep.row_e) { const int2 em = ep.row_e[t]; side6 = em.x; side2 = em.y; }
        }
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
      if constexpr (C::I8 && !C::CERT) {
        if (ep.col_ds) side6 = __float_as_int(ep.col_ds[n]);
      }
    }
  }
  const bool has_out = C::I8 && op.Ao != nullptr;
  [[maybe_unused]] const bool sub_e = C::I8 && !C::CERT && ep.row_e != nullptr;   // wave-uniform
  int lead_ks = 4;
  if (has_out && op.n_out != nullptr) lead_ks = (*op.n_out + 31) >> 5;    // wave-uniform scalar load

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < C::NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;   // all-zero bits: also i32 zero

  i32x4 abl_a[C::MI], abl_b[C::NI];
  if constexpr (C::ABL_NOREAD) {
#pragma unroll
    for (int i = 0; i < C::MI; ++i) abl_a[i] = *reinterpret_cast<const i32x4 *>(op.A + (size_t)(m0 + i * 32 + l31) * op.ldA + kh * 16);
#pragma unroll
    for (int j = 0; j < C::NI; ++j) abl_b[j] = *reinterpret_cast<const i32x4 *>(op.B + (size_t)(n0 + j * 32 + l31) * op.ldB + kh * 16);
  }

  // tile sequence: [outlier tile (int8, optional)] then the nk main k-tiles
  const int lead = has_out ? 1 : 0;
  const int ntiles = op.nk + lead;
  const GemmStageLane sl_main = gemm_stage_lane(lane, (unsigned)op.ldA);   // ldA == ldB (launcher)
  const GemmStageLane sl_lead = gemm_stage_lane(lane, 128u);
  const bool lead_compact = lead_ks <= 1;                  // wave-uniform (device-side count of outlier dims)
  // ONE row of output tiles (T <= BM): all workgroups read the SAME A tile, and walking its k-tiles in step they ask
  // the same few L2 lines at the same moment.  Integer accumulation does not depend on the order, so each workgroup
  // starts its k-walk at its own k-tile (the 32 workgroups of an XCD at 32 different ones) and wraps around.
  const int krot = (C::I8 && nM == 1 && !C::CERT) ? (int)(gridDim.x >= 64 ? blockIdx.x >> 3 : blockIdx.x) % op.nk : 0;   // (few tiles: the sample pass)
  auto stage = [&](int tm0, int tn0, int tile, int slot) {
    if (tile < lead) {
      if constexpr (C::I8) {
        if (lead_compact) { gemm_stage_lead_compact<C>(op.Ao, op.Bo, tm0, tn0, smem, slot, wave, lane); return; }
      }
      gemm_stage<C>(op.Ao, op.Bo, 128, tm0, tn0, 0, smem, slot, wave, sl_lead);
    } else {
      int kq = tile - lead + krot;                         // (a select, no control flow: see gemm_stage_pieces)
      kq -= kq >= op.nk ? op.nk : 0;
      // tile-major: the wave's share is one contiguous block of ONE operand (gemm_stage_block); only that operand's address is formed
      [[maybe_unused]] const bool wa = wave * C::PPW < C::A_PIECES;                 // wave-uniform
      [[maybe_unused]] const unsigned char *wop = wa ? op.A : op.B;
      [[maybe_unused]] const size_t wrt = (size_t)(wa ? tm0 / C::BM : tn0 / C::BN);
      [[maybe_unused]] const size_t wbytes = wa ? (size_t)C::A_BYTES : (size_t)C::B_BYTES;
      [[maybe_unused]] const size_t win = (size_t)(wave * C::PPW - (wa ? 0 : C::A_PIECES)) * 1024;
      [[maybe_unused]] unsigned char *wdst = smem + slot * C::STAGE_BYTES + wave * C::PPW * 1024;
      if constexpr (C::CERT) {                             // wave-uniform; scalar selects (see GemmOperands::cert)
        const int seg = kq >= 2 * op.cert ? 2 : (kq >= op.cert ? 1 : 0), kk = kq - seg * op.cert;
        const int ia = (seg == 0 ? 0 : op.cert) + kk, ib = (seg == 1 ? 0 : op.cert) + kk;
        if constexpr (GemmWaveBlock<C>::OK)
          gemm_stage_block<C>(wop + (wrt * (2 * op.cert) + (wa ? ia : ib)) * wbytes + win, wdst, lane);
        else
          gemm_stage_packed<C>(op.A + ((size_t)(tm0 / C::BM) * (2 * op.cert) + ia) * C::A_BYTES,
                               op.B + ((size_t)(tn0 / C::BN) * (2 * op.cert) + ib) * C::B_BYTES, smem, slot, wave, lane);
        return;
      }
      if (op.packed) {                                     // wave-uniform
        if constexpr (GemmWaveBlock<C>::OK)
          gemm_stage_block<C>(wop + (wrt * op.nk + kq) * wbytes + win, wdst, lane);
        else
          gemm_stage_packed<C>(op.A + ((size_t)(tm0 / C::BM) * op.nk + kq) * C::A_BYTES,
                               op.B + ((size_t)(tn0 / C::BN) * op.nk + kq) * C::B_BYTES, smem, slot, wave, lane);
      } else
        gemm_stage<C>(op.A, op.B, op.ldA, tm0, tn0, (size_t)kq * C::ROWB, smem, slot, wave, sl_main);
    }
  };
  if (tile_id == (int)blockIdx.x) {   // later tiles: staged by their predecessor
    stage(m0, n0, 0, 0);
  }

  // m and -E of the tile's rows are parked EARLY (in front of the tile's first barrier), i.e. while slower waves of this persistent
  // workgroup may still be in the PREVIOUS tile's epilogue: they may only go where that epilogue does not read -- the ROW parts of
  // slots 2 and 6 (its rows' m comes from slot 4, E is not an epilogue input; the COLUMN part of slot 2 is Q_n, read by every
  // gemm_band_sq).  Round 6 first packed the two as int2 across the whole of slot 2: a fast wave's next-tile pairs overwrote Q_n under
  // the slow waves' band computations -- sample features' upper values went wrong at N = 262144 (four sample tiles per workgroup;
  // 0.23 % of the tokens verified-and-wrong in tools/soak_fused.py, all missing features = 13 mod 32), found by the wide soak.
  int *side_m = reinterpret_cast<int *>(smem + C::LDS_RING_BYTES) + 2 * C::NT;
  [[maybe_unused]] int *side_e = reinterpret_cast<int *>(smem + C::LDS_RING_BYTES) + 6 * C::NT;
  static_assert(C::BM + C::BN <= C::NT, "row part | column part of a side slot");
  auto iteration = [&](int kt, bool park_m = false) {
    MSAE_TLK(kt == 8, 0);
    MSAE_TLK(kt == 9, 5);
    wait_vmcnt<0>();               // this wave's pieces of k-tile kt (and the side constants) landed
    if (park_m) {                  // outlier multipliers of the tile's rows -> LDS (read after this k-tile)
      if (tid_ < C::BM) { side_m[tid_] = side2; if constexpr (C::I8 && !C::CERT) side_e[tid_] = -side6; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    MSAE_TLK(kt == 8, 1);
    __builtin_amdgcn_s_barrier();  // ... and everybody's; the other slot is free again
    MSAE_TLK(kt == 8, 2);
    if (kt == 0) MSAE_TL(1);
    if (kt == 1) MSAE_TL(2);
    auto stage_next = [&]() {
      if (kt + 1 < ntiles) stage(m0, n0, kt + 1, (seq + 1) & 1);
      else if (has_next) stage(m0n, n0n, 0, (seq + 1) & 1);
    };
    // (MSAE_GEMM_STAGGER, tuning.h: on by default since round 3 -- -4 % on the main pass together with tile-major operands, two
    // boxes: profiles/r03_ab_stagger_tile_major.txt, r03_ab_ring64_spilling_build.txt; 0 = off, 2 = odd waves)
    // Waves w and w + 4 share a SIMD.  If both issue their LDS-DMA pieces right behind the barrier the SIMD's MFMA pipe idles
    // for that long in every k-tile; so the upper four waves start with MFMAs on the data already in LDS and issue their pieces
    // behind k-step MSAE_GEMM_STAGGER_AT, while their partners -- done issuing -- keep the pipe busy.  ONE copy of the MFMA
    // code: the two roles differ only in which of the two stage_next() call sites is taken.  (Round 3, ~700 cycles of issue per
    // wave: behind k-step 1.  Round 5, one block per wave = ~100 cycles: behind k-step 0 -- the pieces' time to land is what
    // counts now; behind k-step 2 costs 9 %, everybody at the top 2 %: profiles/r05_ab_stagger_at.txt.)
    const bool late = MSAE_GEMM_STAGGER != 0 && (MSAE_GEMM_STAGGER == 2 ? (wave & 1) != 0 : wave >= C::NWAVES / 2) && !park_m && nM > 1;
    if constexpr (!C::ABL_NOSTAGE) {
      if (!late) stage_next();
    }
    MSAE_TLK(kt == 8, 3);
    const unsigned char *sA = smem + (seq & 1) * C::STAGE_BYTES;
    if (park_m) {
      if (lead_compact) { if (lead_ks > 0) { if constexpr (C::I8) gemm_compute_lead_compact<C>(acc, sA, wr, wc, l31, kh); } }
      else gemm_compute_lead<C>(acc, sA, sA + C::A_BYTES, wr, wc, l31, kh, lead_ks);
    }
    else if constexpr (!C::ABL_NOREAD && !C::ABL_NOMFMA && !C::ABL_NOSTAGE)
      gemm_compute_asm<C>(acc, sA, wr, wc, l31, kh, [&](int pos) {
        if (late && pos == MSAE_GEMM_STAGGER_AT) stage_next();
      });
    else gemm_compute<C>(acc, sA, sA + C::A_BYTES, wr, wc, l31, kh, abl_a, abl_b);
    MSAE_TLK(kt == 8, 4);
    ++seq;
  };
  auto scale_by_m = [&]() {
    // |acc| <= 128 * 127 * 127 < 2^21 here and m <= 1040 (quant_x_kernel clamps and hands larger multipliers' tokens to the exact
    // path): the full-rate 24-bit multiply is exact and the product fits int32
    // (v_mul_lo_u32 is quarter rate: 128 of them per lane cost ~2 us per tile)
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wr * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        const int m = side_m[row];
        int ee = 0;
        if constexpr (C::I8 && !C::CERT) ee = side_e[row];     // -E (0 without the subtractive dither): one v_mad_i32_i24 per accumulator either way
#pragma unroll
        for (int j = 0; j < C::NI; ++j) {
          i32x16 v = __builtin_bit_cast(i32x16, acc[i][j]);
          v[e] = __mul24(v[e], m) + ee;
          acc[i][j] = __builtin_bit_cast(f32x16, v);
        }
      }
  };
  int kt0 = 0;
  if constexpr (C::I8) {
    if (has_out) {   // peeled: the outlier dims were quantised at scale m[t]*sx[t]
      iteration(0, true);
      kt0 = 1;
      if (lead_ks > 0 || sub_e) scale_by_m();   // no outlier dim in this batch: the accumulators are still zero (sub_e: they become -E)
    }
  }
  if constexpr (C::CERT) {
    const int cut = 2 * op.cert;                     // (no outlier tile in this mode: kt0 == 0)
    for (int kt = kt0; kt < cut; ++kt) iteration(kt);
    // the low-order segments are done: acc = round(acc / 128), an arithmetic shift (|acc| < 2^31: encode_cert.h)
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        i32x16 v = __builtin_bit_cast(i32x16, acc[i][j]);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (v[e] + 64) >> 7;
        acc[i][j] = __builtin_bit_cast(f32x16, v);
      }
    for (int kt = cut; kt < ntiles; ++kt) iteration(kt);
  } else {
    for (int kt = kt0; kt < ntiles; ++kt) iteration(kt);
  }

  MSAE_TL(3);
  // park the epilogue constants in LDS (side buffer behind the ring)
  float *side = reinterpret_cast<float *>(smem + C::LDS_RING_BYTES);
  side[tid_] = side0;
  side[C::NT + tid_] = side1;
  reinterpret_cast<int *>(side)[2 * C::NT + tid_] = side2;
  side[3 * C::NT + tid_] = side3;
  side[4 * C::NT + tid_] = side4;
  if constexpr (C::I8 && !C::CERT) reinterpret_cast<int *>(side)[6 * C::NT + tid_] = side6;
  if constexpr (!DENSE) {
    // slot 5 of the separable bound, computed only now: the loads it needs had the whole k-loop to land
    float side5 = 0.f;
    if (tid_ < C::BM) {          // B_t: z sigma of this token against the reference feature
      float b2 = side3 * ref0;
      if constexpr (C::I8 || C::F8) {
        const float rz = side1 * side1 * ep.zz12;
        b2 = __builtin_fmaf(rz * side4 * side4, ref2, __builtin_fmaf(rz, ref1, b2));
      }
      side5 = __builtin_sqrtf(b2) * 1.00001f;
    } else if (tid_ < C::BM + C::BN) {   // h_n >= sqrt of every ratio to the reference feature (0/0 counts as 0)
      const float q = __int_as_float(side2);
      float h2 = q / ref0;
      if constexpr (C::I8 || C::F8) {
        h2 = fmaxf(h2, side3 / ref1);
        if (side4 > 0.f) h2 = fmaxf(h2, side4 / ref2);
      }
      side5 = (q > 0.f || side3 > 0.f || side4 > 0.f) ? __builtin_sqrtf(h2) * 1.00001f : 0.f;
    }
    side[5 * C::NT + tid_] = side5;
  }
  // behind the epilogue's first barrier every wave is done with the last k-tile's slot: the next tile's first main
  // k-tile lands there while the epilogue runs (its outlier tile is already in the other slot)
  gemm_epilogue<C, DENSE>(acc, ep, T, m0, n0, wr, wc, lane, smem, side,
                          [] {}, tl_tile);
  MSAE_TL(6);
  }
}

// Host launcher.  Requires Tp % BM == 0 and N % BN == 0 (checked by the caller's plan).
template <class C, bool DENSE>
inline int gemm_launch(const GemmOperands &op, int T, int Tp, int N, const GemmEpilogue &ep, hipStream_t s) {
  if (Tp % C::BM || N % C::BN || op.nk <= 0 || op.ldA != op.ldB || op.ldA % 128 || op.packed > 1) return MSAE_EINVAL;
  if ((op.cert != 0) != C::CERT) return MSAE_EINVAL;
  if (C::CERT && (!C::I8 || op.cert < 0 || op.nk != 3 * op.cert || op.packed != 1 || op.Ao != nullptr)) return MSAE_EINVAL;
  const int nM = Tp / C::BM, nN = N / C::BN;
  auto kern = gemm_kernel<C, DENSE>;
  MSAE_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  static int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 8 ? cus / 8 * 8 : 8;
  }();
  const char *np_env = getenv("MSAE_GEMM_NONPERSISTENT");
  const int per_cu = C::LDS_BYTES > 80 * 1024 ? 1 : 2;
  const int grid = (np_env || nM * nN <= n_cu * per_cu) ? nM * nN : n_cu * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, op, T, Tp, N, nM, nN, ep);
  return (int)hipGetLastError();
}
