// gather_bw.hip -- probe (not part of the product): bandwidth of the re-score kernel's access pattern -- every lane
// streams ONE 16 KiB row with 16-B loads, two batches of 16 loads in flight -- when the rows come from a table that
// fits the Infinity Cache (8192 rows = 134 MB: the token activations a32 of one batch) against a table that does not
// (131072 rows = 2.1 GB: W_enc).  Decides whether a FEATURE-major re-score (W_enc rows once from HBM, the activation
// vector of every (token, feature) pair from the Infinity Cache) can beat the token-major one.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o tools/bin/gather_bw && tools/bin/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LPR>   // lanes per row: 1 (a lane walks its row alone) or 4 (64 B of a row per quad and step)
__global__ __launch_bounds__(64) void gather_kernel(const float *__restrict__ tab, const int *__restrict__ rows, int n_pairs, int d,
                                                    float *__restrict__ out) {
  const int lane = threadIdx.x, rq = lane / LPR, q = lane % LPR;
  const int pair = blockIdx.x * (64 / LPR) + rq;
  const int r = rows[pair < n_pairs ? pair : 0];
  const float *w = tab + (size_t)r * d + 4 * q;
  constexpr int U = 16, B = 4 * U * LPR;
  f32x4 wa[U], wb[U];
  float acc = 0.f;
  auto fetch = [&](f32x4 (&dst)[U], int kk) {
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * LPR * u);
  };
  auto consume = [&](const f32x4 (&src)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) { acc = __builtin_fmaf(src[u][0], 1.0001f, acc); acc = __builtin_fmaf(src[u][1], 1.0001f, acc);
                                  acc = __builtin_fmaf(src[u][2], 1.0001f, acc); acc = __builtin_fmaf(src[u][3], 1.0001f, acc); }
  };
  fetch(wa, 0);
  for (int kk = 0; kk < d; kk += 2 * B) {
    const bool has_b = kk + B < d;
    if (has_b) fetch(wb, kk + B);
    consume(wa);
    if (kk + 2 * B < d) fetch(wa, kk + 2 * B);
    if (has_b) consume(wb);
  }
  if (pair < n_pairs && q == 0) out[pair] = acc;
}

int main(int argc, char **argv) {
  const int d = 4096, n_pairs = 8192 * 45;
  const int sizes[3] = {8192, 32768, 131072};
  float *tab; CK(hipMalloc(&tab, (size_t)131072 * d * 4)); CK(hipMemset(tab, 0, (size_t)131072 * d * 4));
  int *rows; CK(hipMalloc(&rows, n_pairs * 4));
  float *out; CK(hipMalloc(&out, n_pairs * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int order = 0; order < 2; ++order)
    for (int si = 0; si < 3; ++si) {
      const int R = sizes[si];
      std::vector<int> h(n_pairs);
      unsigned long long z = 88172645463325252ull;
      for (int i = 0; i < n_pairs; ++i) {
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        // order 0: random rows (token-major re-score: a token's candidates are random features);
        // order 1: pair i reads row i / 45 -- the 45 pairs of consecutive lanes share a row (what a counting sort by row gives)
        h[i] = order == 0 ? (int)(z % (unsigned)R) : (i / 45) % R;
      }
      CK(hipMemcpy(rows, h.data(), n_pairs * 4, hipMemcpyHostToDevice));
      for (int lpr = 1; lpr <= 4; lpr *= 4) {
        const int grid = (n_pairs + 64 / lpr - 1) / (64 / lpr);
        float best = 1e30f;
        for (int it = 0; it < 5; ++it) {
          CK(hipEventRecord(e0, 0));
          if (lpr == 1) gather_kernel<1><<<grid, 64>>>(tab, rows, n_pairs, d, out);
          else gather_kernel<4><<<grid, 64>>>(tab, rows, n_pairs, d, out);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("%s rows out of a table of %6d rows (%7.1f MB), %d lane(s) per row: %.3f ms  %.2f TB/s\n", order ? "grouped" : "random ", R,
               (double)R * d * 4 / 1e6, lpr, best, (double)n_pairs * d * 4 / (best * 1e-3) / 1e12);
      }
    }
  return 0;
}
