// gemm_sweep.hip -- standalone tuning harness for csrc/gemm_bf16.h (not part of the product).
// Times every tile/pipeline configuration of the candidate-pass GEMM on the BASELINE shape
// (T=8192, d=4096, N=131072, threshold epilogue with nothing emitted) and cross-checks the dense
// epilogue of each configuration against a naive f32 reference on a small problem.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../multimodal-sae_amd/csrc/gemm_mfma.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short *p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    float f = ((float)(z & 0xFFFFFF) / 8388608.0f) - 1.0f;   // uniform [-1, 1)
    p[i] = f32_to_bf16_bits(f);
  }
}
// int8 operands with a chosen distribution (power / clock against switching activity): sigma > 0: round(N(0, sigma)) clamped to
// +-127 (sum of 4 uniforms); mode 1: |.| of it (non-negative); mode 2: multiples of 2 (7-bit values in the 8-bit grid)
__global__ void fill_i8(signed char *p, size_t n, unsigned seed, float sigma, int mode) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    float g = 0.f;
    for (int q = 0; q < 4; ++q) g += (float)((z >> (16 * q)) & 0xFFFF) / 65536.f - 0.5f;   // variance 4/12
    int v = (int)rintf(g * 1.7320508f * sigma);
    v = v > 127 ? 127 : (v < -127 ? -127 : v);
    if (mode == 1) v = v < 0 ? -v : v;
    if (mode == 2) v &= ~1;
    p[i] = (signed char)v;
  }
}
__global__ void ref_dense(const unsigned short *A, const unsigned short *B, int T, int d, int N, float *out) {
  int n = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc += bf16_bits_to_f32(A[(size_t)t * d + k]) * bf16_bits_to_f32(B[(size_t)n * d + k]);
  out[(size_t)t * N + n] = acc;
}
__global__ void max_diff(const float *a, const float *b, size_t n, float *res) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax((int *)res, __float_as_int(m));
}

struct Ctx { unsigned short *A, *B; float *tau, *dense, *ref, *res; f32x4 *rowc, *colc; float *refs; int *cnt; unsigned long long *cand; int T, d, N, Ns; };

template <class C>
void run(const char *name, Ctx &c, int reps) {
  GemmOperands op{};
  op.A = (const unsigned char *)c.A; op.B = (const unsigned char *)c.B;
  op.ldA = op.ldB = (size_t)c.d * 2; op.nk = c.d * 2 / C::ROWB;   // int8 configs read the same bytes as 2*d int8
  GemmEpilogue ep{};
  float md = -1.f;
  if (!C::I8) {  // correctness on the small problem (first Ns features) against the naive reference
    ep.dense = c.dense; ep.ld_dense = c.Ns; ep.bias_stride = 1; ep.rowc = c.rowc; ep.colc = c.colc; ep.refs = c.refs;   // all-zero constants: u = value
    CK(hipMemset(c.dense, 0, (size_t)c.T * c.Ns * 4));
    int rc = gemm_launch<C, true>(op, c.T, c.T, c.Ns, ep, 0);
    if (rc) { printf("%-28s launch failed rc=%d\n", name, rc); return; }
    CK(hipMemset(c.res, 0, 4));
    max_diff<<<1024, 256>>>(c.dense, c.ref, (size_t)512 * c.Ns, c.res);   // ref covers the first 512 tokens
    CK(hipMemcpy(&md, c.res, 4, hipMemcpyDeviceToHost));
  }
  GemmEpilogue et{};
  et.bias_stride = 1; et.tau_vals = c.tau; et.tau_ld = 1; et.tau_col = 0; et.cnt = c.cnt; et.cand = c.cand; et.cap = 16; et.skip_a = et.skip_b = -1;
  et.rowc = c.rowc; et.colc = c.colc; et.refs = c.refs; et.zz12 = 49.f / 12.f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) gemm_launch<C, false>(op, c.T, c.T, c.N, et, 0);
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    gemm_launch<C, false>(op, c.T, c.T, c.N, et, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms;
  }
  double fl = 2.0 * c.T * (C::I8 ? 2.0 * c.d : c.d) * (double)c.N;
  printf("%-28s lds=%3dKB thr=%3d  maxdiff=%.3e  mean %.3f ms (%.0f T%s)  best %.3f ms (%.0f)\n", name, C::LDS_BYTES / 1024, C::NT, md,
         sum / reps, fl / (sum / reps * 1e-3) / 1e12, C::I8 ? "OP/s" : "FLOP/s", best, fl / (best * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv) {
  Ctx c{}; c.T = 8192; c.d = getenv("SWEEP_D") ? atoi(getenv("SWEEP_D")) : 4096; c.N = 131072; c.Ns = 4096;
  printf("d = %d\n", c.d);
  int reps = argc > 1 ? atoi(argv[1]) : 5;
  CK(hipMalloc(&c.A, (size_t)c.T * c.d * 2)); CK(hipMalloc(&c.B, (size_t)c.N * c.d * 2));
  CK(hipMalloc(&c.tau, c.T * 4)); CK(hipMemset(c.tau, 0, c.T * 4));           // tau <= 0 -> nothing emitted
  CK(hipMalloc(&c.rowc, (size_t)c.T * 16)); CK(hipMemset(c.rowc, 0, (size_t)c.T * 16)); CK(hipMalloc(&c.colc, (size_t)c.N * 16)); CK(hipMemset(c.colc, 0, (size_t)c.N * 16));
  { float one[4] = {1.f, 1.f, 1.f, 1.f}; CK(hipMalloc(&c.refs, 16)); CK(hipMemcpy(c.refs, one, 16, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&c.cnt, c.T * 4)); CK(hipMemset(c.cnt, 0, c.T * 4)); CK(hipMalloc(&c.cand, (size_t)c.T * 16 * 8));
  CK(hipMalloc(&c.dense, (size_t)c.T * c.Ns * 4)); CK(hipMalloc(&c.ref, (size_t)512 * c.Ns * 4)); CK(hipMalloc(&c.res, 4));
  if (getenv("SWEEP_ZERO")) { CK(hipMemset(c.A, 0, (size_t)c.T * c.d * 2)); CK(hipMemset(c.B, 0, (size_t)c.N * c.d * 2)); printf("ZERO-FILLED operands\n"); }
  else { fill_bf16<<<4096, 256>>>(c.A, (size_t)c.T * c.d, 1); fill_bf16<<<4096, 256>>>(c.B, (size_t)c.N * c.d, 2); }
  if (getenv("SWEEP_SIGMA")) {   // "sigmaA,sigmaB[,modeA,modeB]"
    float sa = 32.f, sb = 32.f; int ma = 0, mb = 0;
    sscanf(getenv("SWEEP_SIGMA"), "%f,%f,%d,%d", &sa, &sb, &ma, &mb);
    fill_i8<<<4096, 256>>>((signed char *)c.A, (size_t)c.T * c.d * 2, 1, sa, ma);
    fill_i8<<<4096, 256>>>((signed char *)c.B, (size_t)c.N * c.d * 2, 2, sb, mb);
    printf("int8 operands: sigma %.0f / %.0f, mode %d / %d\n", sa, sb, ma, mb);
  }
  ref_dense<<<dim3(c.Ns / 256, 512), 256>>>(c.A, c.B, 512, c.d, c.Ns, c.ref);
  CK(hipDeviceSynchronize());
#define RUN(...) run<GemmCfg<__VA_ARGS__>>(#__VA_ARGS__, c, reps)
  const char *only = getenv("SWEEP_ONLY");   // "pp": just the schedule comparison
  const int which = getenv("SWEEP_CFG") ? atoi(getenv("SWEEP_CFG")) : -1;   // one configuration only (power sampling: tools/smi_sample.sh)
  if (which < 0 || which == 0) RUN(256, 256, 2, 2, 4, false);
  if (which < 0 || which == 1) RUN(256, 256, 2, 2, 4, true);
  if (!only) {
    if (which < 0 || which == 2) RUN(256, 256, 2, 2, 4, false, 24);     // staging only, 8 waves
    if (which < 0 || which == 3) RUN(256, 256, 2, 2, 4, true, 24);      // staging only, int8 byte layout
    if (which < 0 || which == 4) RUN(256, 256, 2, 2, 4, true, 4);       // no staging (MFMA + reads + barriers)
    if (which == 5) RUN(256, 256, 2, 2, 4, true, 8);                    // no fragment reads (staging + MFMA on constant registers)
    if (which == 6) RUN(256, 256, 2, 2, 4, true, 16);                   // no MFMA (staging + fragment reads)
  }
  return 0;
}
