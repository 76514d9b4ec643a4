// gemm96_probe.hip -- timing of the 96-byte / 3-slot ring prototype (tools/tuning/gemm_mfma96.h) against the product's candidate GEMM
// on the bench shape (8192 tokens x 131072 features, 4096 int8 per row; the prototype pads to 43 x 96 = 4128).  Operand content is
// random int8; thresholds are 0, so neither kernel emits candidates (the element loop of the epilogue runs, the pushes do not).
#include <cstdio>
#include <cstdlib>
#include "tuning/gemm_mfma96.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_i8(signed char *p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    float g = 0.f;
    for (int q = 0; q < 4; ++q) g += (float)((z >> (16 * q)) & 0xFFFF) / 65536.f - 0.5f;
    int v = (int)rintf(g * 1.7320508f * 32.f);
    p[i] = (signed char)(v > 127 ? 127 : (v < -127 ? -127 : v));
  }
}

template <class L>
static void timeit(const char *name, L &&launch, int reps, double ops) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) { int rc = launch(); if (rc) { printf("%s: launch rc %d\n", name, rc); return; } }
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms;
  }
  printf("%-52s mean %.3f ms  best %.3f ms  (%.0f TOP/s)\n", name, sum / reps, best, ops / (sum / reps * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int T = 8192, N = 126976, d = 4096, reps = argc > 1 ? atoi(argv[1]) : 10;   // N: the main pass's 31/32 of 131072
  const int nk128 = d / 128, nk96 = (d + 95) / 96;
  const size_t a_bytes = (size_t)(T / 256) * nk96 * 24576, b_bytes = (size_t)(N / 256) * nk96 * 24576;   // >= the 128-B layout's sizes
  unsigned char *A, *B; CK(hipMalloc(&A, a_bytes)); CK(hipMalloc(&B, b_bytes));
  fill_i8<<<4096, 256>>>((signed char *)A, a_bytes, 1); fill_i8<<<4096, 256>>>((signed char *)B, b_bytes, 2);
  float *tau, *refs; f32x4 *rowc, *colc; int *cnt; unsigned long long *cand;
  CK(hipMalloc(&tau, T * 4)); CK(hipMemset(tau, 0, T * 4));
  CK(hipMalloc(&rowc, (size_t)T * 16)); CK(hipMemset(rowc, 0, (size_t)T * 16));
  CK(hipMalloc(&colc, (size_t)N * 16)); CK(hipMemset(colc, 0, (size_t)N * 16));
  { float one[4] = {1.f, 1.f, 1.f, 1.f}; CK(hipMalloc(&refs, 16)); CK(hipMemcpy(refs, one, 16, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&cnt, T * 4)); CK(hipMemset(cnt, 0, T * 4)); CK(hipMalloc(&cand, (size_t)T * 16 * 8));
  CK(hipDeviceSynchronize());
  GemmEpilogue et{};
  et.bias_stride = 1; et.tau_vals = tau; et.tau_ld = 1; et.tau_col = 0; et.cnt = cnt; et.cand = cand; et.cap = 16; et.skip_a = et.skip_b = -1;
  et.rowc = rowc; et.colc = colc; et.refs = refs; et.zz12 = 49.f / 12.f;
  GemmOperands o128{}; o128.A = A; o128.B = B; o128.ldA = o128.ldB = (size_t)d; o128.nk = nk128; o128.packed = 1;
  GemmOperands o96{};  o96.A = A;  o96.B = B;  o96.ldA = o96.ldB = (size_t)nk96 * 96; o96.nk = nk96; o96.packed = 1;
  using P = GemmCfg<256, 256, 2, 2, 4, true>;
  const double ops = 2.0 * T * (double)d * N;
  for (int rep = 0; rep < 2; ++rep) {
    timeit("product: 128-B k-tiles x 2 slots (32 k-tiles)", [&] { return gemm_launch<P, false>(o128, T, T, N, et, 0); }, reps, ops);
    timeit("prototype: 96-B k-tiles x 3 slots (43 k-tiles), late @0", [&] { return gemm96_launch<Gemm96Cfg, 0>(o96, T, T, N, et, 0); }, reps, ops);
    timeit("prototype: 96-B k-tiles x 3 slots (43 k-tiles), late @1", [&] { return gemm96_launch<Gemm96Cfg, 1>(o96, T, T, N, et, 0); }, reps, ops);
  }
  return 0;
}
