// dma_depth.hip -- standalone tuning probe (not part of the product): L2 -> LDS streaming throughput of the candidate
// GEMM's operand pattern (persistent workgroup per CU, XCD-aware 8x4 super-tiles, int8 operands of T=8192 / N=131072 /
// d=4096) as a function of the k-tile width and the ring depth, WITHOUT any MFMA or fragment reads.  Answers: would a
// deeper ring of narrower k-tiles (more bytes in flight per CU) deliver the tiles faster than the 2 x 64 KB ring?
#include <cstdio>
#include <cstdlib>
#include "../multimodal-sae_amd/csrc/gemm_mfma.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BK, int DEPTH>
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
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(smem + slot * SLOT + piece * 1024), 16, 0, 0);
    }
  };
  const int tiles = (nM * nN - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // output tiles of this workgroup
  const long total = (long)tiles * nk;                                                   // its flat k-tile sequence
  auto stage_seq = [&](long i) {
    if (i < total) stage((int)blockIdx.x + (int)(i / nk) * (int)gridDim.x, (int)(i % nk), (int)(i % DEPTH));
  };
  for (int i = 0; i < DEPTH - 1; ++i) stage_seq(i);
  for (long i = 0; i < total; ++i) {
    vmcnt_le<(DEPTH - 2) * PPW>();          // k-tile i landed (this wave's pieces); the newer ones may be in flight
    __builtin_amdgcn_s_barrier();
    stage_seq(i + DEPTH - 1);
  }
  vmcnt_le<0>();
}

template <int BK, int DEPTH>
void run(const unsigned char *A, const unsigned char *B, int T, int N, int d, int reps) {
  const int nM = T / 256, nN = N / 256, nk = d / BK;
  const size_t smem = (size_t)DEPTH * 512 * BK;
  auto kern = stream_kernel<BK, DEPTH>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, 0, A, B, (size_t)d, nM, nN, nk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bytes = (double)nM * nN * nk * 512.0 * BK;
  printf("k-tile %3d B x ring %d (%3zu KB LDS, %3d KB in flight): %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s  = %5.2f us per 64 KB\n", BK, DEPTH,
         smem >> 10, (DEPTH - 1) * 512 * BK >> 10, best, bytes / best / 1e6 / 256, bytes / best / 1e9, best * 1e3 / ((double)nM * nN * d / 128 / 256));
}

int main() {
  const int T = 8192, N = 131072, d = 4096;
  unsigned char *A, *B;
  CK(hipMalloc(&A, (size_t)T * d)); CK(hipMalloc(&B, (size_t)N * d));
  CK(hipMemset(A, 1, (size_t)T * d)); CK(hipMemset(B, 2, (size_t)N * d));
  run<128, 2>(A, B, T, N, d, 5);
  run<64, 2>(A, B, T, N, d, 5);
  run<64, 3>(A, B, T, N, d, 5);
  run<64, 4>(A, B, T, N, d, 5);
  run<32, 4>(A, B, T, N, d, 5);
  run<32, 8>(A, B, T, N, d, 5);
  run<128, 2>(A, B, T, N, d, 5);
  return 0;
}
