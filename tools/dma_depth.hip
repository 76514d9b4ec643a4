// dma_depth.hip -- standalone tuning probe (not part of the product): L2 -> LDS streaming throughput of the candidate
// GEMM's operand pattern (persistent workgroup per CU, XCD-aware 8x4 super-tiles, int8 operands of T=8192 / N=131072 /
// d=4096) as a function of the k-tile width and the ring depth, WITHOUT any MFMA or fragment reads.  Answers: would a
// deeper ring of narrower k-tiles (more bytes in flight per CU) deliver the tiles faster than the 2 x 64 KB ring?
#include <cstdio>
#include <cstdlib>
#include "../multimodal-sae_amd/csrc/gemm_mfma.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PACKED (round 3): tile-major operands -- the BK-byte k-tile of a 256-row tile is ONE contiguous 256 x BK block
// ([row tile][k-tile][256][BK]), so a 1-KiB piece is 8 consecutive 128-B lines instead of 8 lines 4 KB apart.
template <int BK, int DEPTH, bool COMPUTE = false, bool PACKED = false>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char *__restrict__ A, const unsigned char *__restrict__ B,
                                                     size_t ld, int nM, int nN, int nk) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int ROWS_PER_PIECE = 1024 / BK, LANES_PER_ROW = BK / 16;
  constexpr int PIECES = 512 * BK / 1024, PPW = PIECES / 8, SLOT = 512 * BK;
  static_assert(PPW >= 1 && DEPTH >= 2 && DEPTH * SLOT <= 160 * 1024 - 4096, "ring");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane / LANES_PER_ROW, lcol = (lane % LANES_PER_ROW) * 16;
  auto stage = [&](int tile_id, int kt, int slot) {
    int tm, tn;
    gemm_map_tile(tile_id, nM, nN, tm, tn);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i;                  // 0 .. PIECES-1: first half A rows, second half B rows
      const bool isA = piece < PIECES / 2;
      const int pl = isA ? piece : piece - PIECES / 2;
      const unsigned char *src = (isA ? A + (size_t)(tm * 256 + pl * ROWS_PER_PIECE + lrow) * ld
                                      : B + (size_t)(tn * 256 + pl * ROWS_PER_PIECE + lrow) * ld) + (size_t)kt * BK + lcol;
      if constexpr (PACKED)
        src = (isA ? A + ((size_t)tm * nk + kt) * (256 * BK) : B + ((size_t)tn * nk + kt) * (256 * BK)) + pl * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(smem + slot * SLOT + piece * 1024), 16, 0, 0);
    }
  };
  const int tiles = (nM * nN - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // output tiles of this workgroup
  const long total = (long)tiles * nk;                                                   // its flat k-tile sequence
  auto stage_seq = [&](long i) {
    if (i < total) stage((int)blockIdx.x + (int)(i / nk) * (int)gridDim.x, (int)(i % nk), (int)(i % DEPTH));
  };
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  typedef int v16i_ __attribute__((ext_vector_type(16)));
  v16i_ acc[8];
  if constexpr (COMPUTE) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0;
  }
  for (int i = 0; i < DEPTH - 1; ++i) stage_seq(i);
  for (long i = 0; i < total; ++i) {
    vmcnt_le<(DEPTH - 2) * PPW>();          // k-tile i landed (this wave's pieces); the newer ones may be in flight
    __builtin_amdgcn_s_barrier();
    stage_seq(i + DEPTH - 1);
    if constexpr (COMPUTE) {                // the product's work per 32 bytes of k: 6 fragment reads + 8 MFMAs (128 x 64 per wave)
      const unsigned char *slot = smem + (int)(i % DEPTH) * SLOT;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        v4i_ f[6];
#pragma unroll
        for (int q = 0; q < 6; ++q)
          f[q] = *reinterpret_cast<const v4i_ *>(slot + (((wave * 6 + q) * 32 + (lane & 31)) * BK) % SLOT + (((ks * 2 + (lane >> 5)) * 16) % BK));
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[q & 3], f[4 + (q >> 2)], acc[q], 0, 0, 0);
      }
    }
  }
  vmcnt_le<0>();
  if constexpr (COMPUTE) {
    int sx = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sx ^= acc[q][0] ^ acc[q][15];
    if (sx == 0x7fffffff) smem[0] = 1;
  }
}

// Register path for comparison: the same pieces through global_load_dwordx4 -> VGPR (-> ds_write_b128 when WRITE),
// one k-tile (PPW x 16 B per lane) in flight while the previous one is written / dropped.
template <int BK, bool WRITE, bool PACKED = false>
__global__ __launch_bounds__(512) void stream_reg_kernel(const unsigned char *__restrict__ A, const unsigned char *__restrict__ B,
                                                         size_t ld, int nM, int nN, int nk, int *sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int ROWS_PER_PIECE = 1024 / BK, LANES_PER_ROW = BK / 16;
  constexpr int PIECES = 512 * BK / 1024, PPW = PIECES / 8, SLOT = 512 * BK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane / LANES_PER_ROW, lcol = (lane % LANES_PER_ROW) * 16;
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  auto fetch = [&](long i, i32x4v (&r)[PPW]) {
    const int tile_id = (int)blockIdx.x + (int)(i / nk) * (int)gridDim.x, kt = (int)(i % nk);
    int tm, tn;
    gemm_map_tile(tile_id, nM, nN, tm, tn);
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      const int piece = wave * PPW + p;
      const bool isA = piece < PIECES / 2;
      const int pl = isA ? piece : piece - PIECES / 2;
      const unsigned char *src = (isA ? A + (size_t)(tm * 256 + pl * ROWS_PER_PIECE + lrow) * ld
                                      : B + (size_t)(tn * 256 + pl * ROWS_PER_PIECE + lrow) * ld) + (size_t)kt * BK + lcol;
      if constexpr (PACKED)
        src = (isA ? A + ((size_t)tm * nk + kt) * (256 * BK) : B + ((size_t)tn * nk + kt) * (256 * BK)) + pl * 1024 + lane * 16;
      r[p] = *reinterpret_cast<const i32x4v *>(src);
    }
  };
  const int tiles = (nM * nN - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const long total = (long)tiles * nk;
  i32x4v ra[PPW], rb[PPW];
  int acc = 0;
  if (total > 0) fetch(0, ra);
  for (long i = 0; i < total; i += 2) {
    if (i + 1 < total) fetch(i + 1, rb);
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      if (WRITE) *reinterpret_cast<i32x4v *>(smem + (i & 1) * SLOT + (wave * PPW + p) * 1024 + lane * 16) = ra[p];
      else acc ^= ra[p][0] ^ ra[p][3];
    }
    if (WRITE) __builtin_amdgcn_s_barrier();
    if (i + 2 < total) fetch(i + 2, ra);
    if (i + 1 < total) {
#pragma unroll
      for (int p = 0; p < PPW; ++p) {
        if (WRITE) *reinterpret_cast<i32x4v *>(smem + ((i + 1) & 1) * SLOT + (wave * PPW + p) * 1024 + lane * 16) = rb[p];
        else acc ^= rb[p][1] ^ rb[p][2];
      }
      if (WRITE) __builtin_amdgcn_s_barrier();
    }
  }
  if (acc == 0x12345678) sink[0] = acc;
}

static int g_grid = 256;
template <int BK, int DEPTH, bool COMPUTE = false, bool PACKED = false>
void run(const unsigned char *A, const unsigned char *B, int T, int N, int d, int reps) {
  const int nM = T / 256, nN = N / 256, nk = d / BK;
  const size_t smem = (size_t)DEPTH * 512 * BK;
  auto kern = stream_kernel<BK, DEPTH, COMPUTE, PACKED>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(g_grid), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(g_grid), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bytes = (double)nM * nN * nk * 512.0 * BK;
  printf("%s%sk-tile %3d B x ring %d (%3zu KB LDS, %3d KB in flight): %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s  = %5.2f us per 64 KB\n",
         PACKED ? "tile-major " : "row-major  ", COMPUTE ? "+reads+MFMA " : "", BK, DEPTH,
         smem >> 10, (DEPTH - 1) * 512 * BK >> 10, best, bytes / best / 1e6 / g_grid, bytes / best / 1e9, best * 1e3 / ((double)nM * nN * d / 128 / g_grid));
}

template <int BK, bool WRITE, bool PACKED = false>
void run_reg(const unsigned char *A, const unsigned char *B, int T, int N, int d, int reps, int *sink) {
  const int nM = T / 256, nN = N / 256, nk = d / BK;
  const size_t smem = (size_t)2 * 512 * BK;
  auto kern = stream_reg_kernel<BK, WRITE, PACKED>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(g_grid), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk, sink);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(g_grid), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bytes = (double)nM * nN * nk * 512.0 * BK;
  printf("%sk-tile %3d B via VGPRs%s: %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s  = %5.2f us per 64 KB\n",
         PACKED ? "tile-major " : "row-major  ", BK, WRITE ? " + ds_write_b128 + barrier" : " (dropped)             ", best, bytes / best / 1e6 / g_grid, bytes / best / 1e9,
         best * 1e3 / ((double)nM * nN * d / 128 / g_grid));
}

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 131072, d = 4096;
  if (argc > 3) g_grid = atoi(argv[3]);
  printf("T=%d N=%d d=%d: x %d MB, Wq %d MB, %d workgroups\n", T, N, d, T / 256, N / 256, g_grid);
  unsigned char *A, *B;
  CK(hipMalloc(&A, (size_t)T * d)); CK(hipMalloc(&B, (size_t)N * d));
  CK(hipMemset(A, 1, (size_t)T * d)); CK(hipMemset(B, 2, (size_t)N * d));
  if (argc > 4 && atoi(argv[4]) == 1) {      // random bytes instead of constants (DVFS: constant operands clock higher)
    unsigned char *h = (unsigned char *)malloc((size_t)N * d);
    unsigned x = 12345u;
    for (size_t i = 0; i < (size_t)N * d; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned char)(x >> 24); }
    CK(hipMemcpy(B, h, (size_t)N * d, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, h, (size_t)T * d, hipMemcpyHostToDevice));
    free(h);
  }
  int *sink; CK(hipMalloc(&sink, 64));
  for (int rep = 0; rep < 2; ++rep) {         // A/B interleaved, twice
    run<128, 2>(A, B, T, N, d, 5);
    run<128, 2, false, true>(A, B, T, N, d, 5);
    run<64, 4>(A, B, T, N, d, 5);
    run<64, 4, false, true>(A, B, T, N, d, 5);
    run<32, 8>(A, B, T, N, d, 5);
    run<32, 8, false, true>(A, B, T, N, d, 5);
    run<128, 2, true>(A, B, T, N, d, 5);
    run<128, 2, true, true>(A, B, T, N, d, 5);
    run<64, 4, true>(A, B, T, N, d, 5);
    run<64, 4, true, true>(A, B, T, N, d, 5);
    run_reg<128, false>(A, B, T, N, d, 5, sink);
    run_reg<128, false, true>(A, B, T, N, d, 5, sink);
    run_reg<128, true>(A, B, T, N, d, 5, sink);
    run_reg<128, true, true>(A, B, T, N, d, 5, sink);
  }
  return 0;
}
