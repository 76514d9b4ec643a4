// fm_rescore_probe.hip -- probe (not part of the product): a FEATURE-major exact re-score.  The (token, feature) pairs of one
// batch sorted by feature; lane i owns pair i and walks BOTH rows itself: its feature's row of W_enc (f32, 16 KB; consecutive
// lanes mostly share it -- the texture path merges identical addresses, HBM sees each row once) and its token's activation row
// in the caller's bf16 (8 KB, out of the 67 MB table the Infinity Cache holds), centred on the fly (a = float(x) - b_dec, the
// reference's own expression) and chained in ascending k like the product's kernel.  Against the token-major kernel's cost of
// 16 KB of HBM per pair (select_rescore_kernel: 1.03 ms at k = 32 / 45 rows per token, 7.98 ms at k = 256 / 347).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fm_rescore_probe.hip -o tools/bin/fm_rescore_probe && tools/bin/fm_rescore_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>   // U 16-B pieces of W per lane and batch (two batches in flight); NT: non-temporal W loads
__global__ __launch_bounds__(64) void fm_kernel(const float *__restrict__ W, const unsigned short *__restrict__ x,
                                                const float *__restrict__ b_dec, const int2 *__restrict__ pairs, int n_pairs, int d,
                                                float *__restrict__ out) {
  const int lane = threadIdx.x, pair = blockIdx.x * 64 + lane;
  const int2 ft = pairs[pair < n_pairs ? pair : n_pairs - 1];
  const float *__restrict__ w = W + (size_t)ft.x * d;
  const unsigned short *__restrict__ xr = x + (size_t)ft.y * d;
  constexpr int B = 4 * U;                  // floats per batch
  f32x4 wa[U], wb[U];
  u32x4 xa[U / 2], xb[U / 2];
  float acc = 0.f;
  auto fetch = [&](f32x4 (&dw)[U], u32x4 (&dx)[U / 2], int kk) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (NT) dw[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(w + kk + 4 * u));
      else dw[u] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < U / 2; ++u) dx[u] = *reinterpret_cast<const u32x4 *>(xr + kk + 8 * u);
  };
  auto consume = [&](const f32x4 (&sw)[U], const u32x4 (&sx)[U / 2], int kk) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned pk = sx[u >> 1][(u & 1) * 2 + (e >> 1)];
        const float xf = __uint_as_float((e & 1) ? (pk & 0xFFFF0000u) : (pk << 16));
        const float a = xf - b_dec[kk + 4 * u + e];        // wave-uniform address: scalar load
        acc = __builtin_fmaf(a, sw[u][e], acc);
      }
    }
  };
  fetch(wa, xa, 0);
  for (int kk = 0; kk < d; kk += 2 * B) {
    fetch(wb, xb, kk + B);
    consume(wa, xa, kk);
    if (kk + 2 * B < d) fetch(wa, xa, kk + 2 * B);
    consume(wb, xb, kk + B);
  }
  if (pair < n_pairs) out[pair] = acc;
}

int main() {
  const int d = 4096, N = 131072, T = 8192;
  float *W; CK(hipMalloc(&W, (size_t)N * d * 4)); CK(hipMemset(W, 0, (size_t)N * d * 4));
  unsigned short *x; CK(hipMalloc(&x, (size_t)T * d * 2)); CK(hipMemset(x, 0, (size_t)T * d * 2));
  float *bd; CK(hipMalloc(&bd, d * 4)); CK(hipMemset(bd, 0, d * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rpts[3] = {45, 120, 347};      // rows per token: k = 32, (k = 96), k = 256
  for (int ri = 0; ri < 3; ++ri) {
    const int n_pairs = T * rpts[ri];
    std::vector<int2> h(n_pairs);
    unsigned long long z = 88172645463325252ull;
    for (int sorted = 1; sorted >= 0; --sorted) {
      for (int i = 0; i < n_pairs; ++i) {
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        h[i].x = (int)(z % (unsigned)N);
        h[i].y = sorted ? (int)((z >> 32) % (unsigned)T) : i / rpts[ri];   // unsorted = the token-major order (consecutive lanes share the TOKEN)
      }
      if (sorted) std::sort(h.begin(), h.end(), [](const int2 &a, const int2 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
      int2 *pairs; CK(hipMalloc(&pairs, (size_t)n_pairs * 8)); CK(hipMemcpy(pairs, h.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice));
      float *out; CK(hipMalloc(&out, (size_t)n_pairs * 4));
      for (int var = 0; var < 3; ++var) {
        float best = 1e30f;
        const int grid = (n_pairs + 63) / 64;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(e0, 0));
          if (var == 0) fm_kernel<16, false><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          else if (var == 1) fm_kernel<16, true><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          else fm_kernel<8, true><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("%3d rows/token (%7d pairs, %4.1f per feature) %s  U=%2d %s W loads: %.3f ms\n", rpts[ri], n_pairs, (double)n_pairs / N,
               sorted ? "sorted by feature" : "token-major order", var == 2 ? 8 : 16, var ? "non-temporal" : "default     ", best);
      }
      CK(hipFree(pairs)); CK(hipFree(out));
    }
  }
  return 0;
}
