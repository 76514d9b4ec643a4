// mfma_rate.hip -- probe (not part of the product): sustained matrix rate of one MFMA flavour with every SIMD busy and nothing
// else in the loop (8 independent accumulators per wave, 2 waves per SIMD), i.e. the rate the package power cap allows.
// int8 32x32x32 against the f8f6f4 32x32x64 instruction with fp8 / fp6 / fp4 operands (would an MXFP6
// candidate pass really run twice as fast as the int8 one?).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FMT>   // -1: int8 32x32x32;  0 fp8 (e4m3), 2 fp6 (e2m3), 4 fp4 (e2m1) through v_mfma_scale_f32_32x32x64_f8f6f4
__global__ __launch_bounds__(512) void rate_kernel(const int *__restrict__ src, float *__restrict__ out, int iters, int rot) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  i32x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 8 + i) & 0xFFFF]; b[i] = src[(tid * 8 + i + 4096) & 0xFFFF]; }
  if constexpr (FMT < 0) {
    i32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = i32x16{};
    // MFMA_RATE_VARY (rot != 0): consecutive MFMAs read DIFFERENT operand registers (4 sets), as a GEMM's do; otherwise all of
    // them read the same two registers and only the accumulators toggle
    i32x4 a4[4], b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = rot ? q : 0;
      a4[q] = i32x4{a[o & 7], a[(o + 1) & 7], a[(o + 2) & 7], a[(o + 3) & 7]};
      b4[q] = i32x4{b[(o + 4) & 7], b[(o + 5) & 7], b[(o + 6) & 7], b[(o + 7) & 7]};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4[j & 3], b4[j & 3], acc[j], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][7];
    out[tid] = (float)s;
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x16{};
    i32x8 av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = rot ? 2 * q : 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) { av[q][e] = a[(e + o) & 7]; bv[q][e] = b[(e + o + 1) & 7]; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[j & 3], bv[j & 3], acc[j], FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
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
    rate_kernel<FMT><<<grid, 512>>>(src, out, iters, getenv("MFMA_RATE_VARY") ? 1 : 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
  }
  const double ops = (double)grid * 8 * iters * 8 * ops_per_mfma;
  printf("%-34s %8.3f ms  %6.2f P(FL)OP/s\n", name, best, ops / (best * 1e-3) / 1e15);
}

int main() {
  int *src; CK(hipMalloc(&src, 65536 * 4));
  // small magnitudes: finite in every format.  MFMA_RATE_SIGNED=i8: random signs in two's complement (what the int8 pass sees);
  // MFMA_RATE_SIGNED=fp: random sign BITS (sign-magnitude: what an fp8 / fp6 / fp4 pass would see)
  const char *sg = getenv("MFMA_RATE_SIGNED");
  // MFMA_RATE_GAUSS=i8: round(N(0, 32)) in two's complement -- the int8 pass's operands (per-row scale max / 127, max ~ 4 sigma);
  // MFMA_RATE_GAUSS=fp8: the e4m3 encoding of N(0, 1) * 448 / 4 -- the same data as an MX-scaled e4m3 pass would hold it (block
  // maximum near the top binade, three random mantissa bits, random sign bit).  Round 4: the power-capped MFMA rate on the
  // PRODUCT's operand statistics, the ceiling of what an fp8 candidate GEMM could gain (tools/gpu_r04_power.sh).
  const char *gs = getenv("MFMA_RATE_GAUSS");
  { unsigned char *h = (unsigned char *)malloc(65536 * 4); unsigned z = 12345u;
    for (int i = 0; i < 65536 * 4; ++i) {
      z = z * 1664525u + 1013904223u;
      const unsigned m = (z >> 8) & 0x3Fu, sign = (z >> 20) & 1u;
      h[i] = !sg ? (unsigned char)m : (sg[0] == 'i' ? (unsigned char)(sign ? (unsigned char)(0u - m) : m) : (unsigned char)(sign << 7 | m));
      if (gs) {
        float g = 0.f;
        for (int q = 0; q < 12; ++q) { z = z * 1664525u + 1013904223u; g += (float)(z >> 8) / 16777216.f; }
        g -= 6.f;                                                            // ~N(0, 1)
        if (gs[0] == 'i') {
          int v = (int)rintf(g * 32.f);
          h[i] = (unsigned char)(signed char)(v > 127 ? 127 : (v < -127 ? -127 : v));
        } else {                                                             // e4m3: bias 7, max 448, subnormals below 2^-6
          float a = fabsf(g) * 112.f;
          if (a > 448.f) a = 448.f;
          int e = 0; float mant = 0.f;
          if (a >= 0.015625f) { e = (int)floorf(log2f(a)); if (e > 8) e = 8; mant = a / exp2f((float)e) - 1.f; }
          int mi = (int)rintf(mant * 8.f), eb = e + 7;
          if (a < 0.015625f) { eb = 0; mi = (int)rintf(a / 0.001953125f); }
          if (mi == 8) { mi = 0; ++eb; }
          if (eb > 15 || (eb == 15 && mi > 6)) { eb = 15; mi = 6; }
          h[i] = (unsigned char)((g < 0.f ? 0x80 : 0) | (eb << 3) | mi);
        }
      }
    }
    CK(hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice)); free(h); }
  if (gs) printf("operands: Gaussian, %s\n", gs[0] == 'i' ? "int8 round(N(0, 32))" : "e4m3 of N(0, 1) * 112");
  else if (sg) printf("operands: random signs, %s\n", sg[0] == 'i' ? "two's complement" : "sign bit");
  float *out; CK(hipMalloc(&out, 256 * 512 * 4));
  if (getenv("MFMA_RATE_ONLY_FP8")) { run<0>("f8f6f4 32x32x64, fp8 e4m3", src, out, 2.0 * 32 * 32 * 64); return 0; }
  run<-1>("int8 32x32x32", src, out, 2.0 * 32 * 32 * 32);
  if (getenv("MFMA_RATE_ONLY_I8")) return 0;
  run<0>("f8f6f4 32x32x64, fp8 e4m3", src, out, 2.0 * 32 * 32 * 64);
  run<2>("f8f6f4 32x32x64, fp6 e2m3", src, out, 2.0 * 32 * 32 * 64);
  run<4>("f8f6f4 32x32x64, fp4 e2m1", src, out, 2.0 * 32 * 32 * 64);
  run<-1>("int8 32x32x32 (again)", src, out, 2.0 * 32 * 32 * 32);
  return 0;
}
