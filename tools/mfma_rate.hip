// mfma_rate.hip -- probe (not part of the product): sustained matrix rate of one MFMA flavour with every SIMD busy and nothing
// else in the loop (8 independent accumulators per wave, 2 waves per SIMD), i.e. the rate the package power cap allows.
// int8 32x32x32 against the f8f6f4 32x32x64 instruction with fp8 / fp6 / fp4 operands (would an MXFP6
// candidate pass really run twice as fast as the int8 one?).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FMT>   // -1: int8 32x32x32;  0 fp8 (e4m3), 2 fp6 (e2m3), 4 fp4 (e2m1) through v_mfma_scale_f32_32x32x64_f8f6f4
__global__ __launch_bounds__(512) void rate_kernel(const int *__restrict__ src, float *__restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  i32x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 8 + i) & 0xFFFF]; b[i] = src[(tid * 8 + i + 4096) & 0xFFFF]; }
  if constexpr (FMT < 0) {
    i32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = i32x16{};
    const i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acc[j], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][7];
    out[tid] = (float)s;
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x16{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][7];
    out[tid] = s;
  }
}

template <int FMT>
void run(const char *name, const int *src, float *out, double ops_per_mfma) {
  const int grid = 256, iters = 20000;    // one 8-wave workgroup per CU: 2 waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  const int reps = getenv("MFMA_RATE_REPS") ? atoi(getenv("MFMA_RATE_REPS")) : 4;   // many: long enough for the SMU's power average
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(e0, 0));
    rate_kernel<FMT><<<grid, 512>>>(src, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
  }
  const double ops = (double)grid * 8 * iters * 8 * ops_per_mfma;
  printf("%-34s %8.3f ms  %6.2f P(FL)OP/s\n", name, best, ops / (best * 1e-3) / 1e15);
}

int main() {
  int *src; CK(hipMalloc(&src, 65536 * 4));
  { int *h = (int *)malloc(65536 * 4); unsigned z = 12345u; for (int i = 0; i < 65536; ++i) { z = z * 1664525u + 1013904223u; h[i] = (int)(z & 0x3F3F3F3Fu); }   // small magnitudes: finite in every format
    CK(hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice)); free(h); }
  float *out; CK(hipMalloc(&out, 256 * 512 * 4));
  run<-1>("int8 32x32x32", src, out, 2.0 * 32 * 32 * 32);
  if (getenv("MFMA_RATE_ONLY_I8")) return 0;
  run<0>("f8f6f4 32x32x64, fp8 e4m3", src, out, 2.0 * 32 * 32 * 64);
  run<2>("f8f6f4 32x32x64, fp6 e2m3", src, out, 2.0 * 32 * 32 * 64);
  run<4>("f8f6f4 32x32x64, fp4 e2m1", src, out, 2.0 * 32 * 32 * 64);
  run<-1>("int8 32x32x32 (again)", src, out, 2.0 * 32 * 32 * 32);
  return 0;
}
