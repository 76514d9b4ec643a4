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

template <int U, int MODE>   // U 16-B pieces of W per lane and batch (two batches in flight); MODE 0: both rows, 1: x rows only, 2: W rows only
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
      if constexpr (MODE != 1) dw[u] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * u);
      else dw[u] = f32x4{1.f, 2.f, 3.f, (float)kk};
    }
#pragma unroll
    for (int u = 0; u < U / 2; ++u) {
      if constexpr (MODE != 2) dx[u] = *reinterpret_cast<const u32x4 *>(xr + kk + 8 * u);
      else dx[u] = u32x4{1u, 2u, 3u, (unsigned)kk};
    }
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

// Grouped variant: the pairs of a feature occupy whole groups of G lanes (padded; pad slots carry the feature and token -1); the
// G lanes of a group load 16 G contiguous bytes of the row of W_enc per instruction -- 1 / G of the row per lane -- and every
// lane takes the element it needs from its neighbour's register through DPP (row_share / quad_perm) inside the fma chain.
// acc = fma(a, w of lane SH of this lane's group, acc): one v_fmac_f32 with the DPP source modifier (the compiler does not fold
// a v_mov_dpp into the fma and hoists 64 of them per batch instead: 128 VGPRs more)
template <int G, int SH>
__device__ __forceinline__ void fma_share(float &acc, float a, float w) {
  if constexpr (G == 16) asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a), "n"(SH));
  else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a), "n"(SH));
}
template <int G, int P, int N>
struct Chain {
  template <class WR, class XR>
  static __device__ __forceinline__ void run(float &acc, const WR &sw, const XR &sx, const float *__restrict__ bd) {
    if constexpr (N < G) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        constexpr int dummy = 0; (void)dummy;
        const int e = (P * G + N) * 4 + c;                 // element of the batch
        const unsigned pk = sx[e >> 3][(e >> 1) & 3];
        const float xf = __uint_as_float((e & 1) ? (pk & 0xFFFF0000u) : (pk << 16));
        const float a = xf - bd[e];
        fma_share<G, N>(acc, a, sw[P][c]);
      }
      Chain<G, P, N + 1>::run(acc, sw, sx, bd);
    }
  }
};
template <int G, int KB, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void fmg_kernel(const float *__restrict__ W, const unsigned short *__restrict__ x,
                                                 const float *__restrict__ b_dec, const int2 *__restrict__ slots, int n_slots, int d,
                                                 float *__restrict__ out) {
  const int lane = threadIdx.x, slot = blockIdx.x * 64 + lane;
  const int2 ft = slots[slot < n_slots ? slot : n_slots - 1];
  const bool valid = slot < n_slots && ft.y >= 0;
  constexpr int WP = KB / (4 * G), XP = KB / 8;    // floats of k per batch; 16-B pieces of W / of x per lane and batch
  const float *__restrict__ w = W + (size_t)ft.x * d + 4 * (lane % G);
  const unsigned short *__restrict__ xr = x + (size_t)(valid ? ft.y : 0) * d;
  f32x4 wa[WP], wb[WP];
  u32x4 xa[XP], xb[XP];
  float acc = 0.f;
  auto fetch = [&](f32x4 (&dw)[WP], u32x4 (&dx)[XP], int kk) {
#pragma unroll
    for (int p = 0; p < WP; ++p) dw[p] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * G * p);
    if (valid) {
#pragma unroll
      for (int u = 0; u < XP; ++u) dx[u] = *reinterpret_cast<const u32x4 *>(xr + kk + 8 * u);
    }
  };
  auto consume = [&](const f32x4 (&sw)[WP], const u32x4 (&sx)[XP], int kk) {
    const float *__restrict__ bd = b_dec + kk;
    if constexpr (WP >= 1) Chain<G, 0, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 2) Chain<G, 1, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 3) Chain<G, 2, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 4) Chain<G, 3, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 5) Chain<G, 4, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 6) Chain<G, 5, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 7) Chain<G, 6, 0>::run(acc, sw, sx, bd);
    if constexpr (WP >= 8) Chain<G, 7, 0>::run(acc, sw, sx, bd);
  };
  fetch(wa, xa, 0);
  for (int kk = 0; kk < d; kk += 2 * KB) {
    fetch(wb, xb, kk + KB);
    consume(wa, xa, kk);
    if (kk + 2 * KB < d) fetch(wa, xa, kk + 2 * KB);
    consume(wb, xb, kk + KB);
  }
  if (valid) out[slot] = acc;
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
          if (var == 0) fm_kernel<16, 0><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          else if (var == 1) fm_kernel<16, 1><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          else fm_kernel<16, 2><<<grid, 64>>>(W, x, bd, pairs, n_pairs, d, out);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("%3d rows/token (%7d pairs, %4.1f per feature) %s  %s: %.3f ms\n", rpts[ri], n_pairs, (double)n_pairs / N,
               sorted ? "sorted by feature" : "token-major order", var == 0 ? "both rows  " : (var == 1 ? "x rows only" : "W rows only"), best);
      }
      if (sorted) {
        for (int G = 4; G <= 16; G *= 4) {
          std::vector<int2> hs;
          hs.reserve(n_pairs + (size_t)N * G);
          for (int i = 0; i < n_pairs;) {
            int j = i;
            while (j < n_pairs && h[j].x == h[i].x) { hs.push_back(h[j]); ++j; }
            while (hs.size() % G) hs.push_back(int2{h[i].x, -1});
            i = j;
          }
          const int n_slots = (int)hs.size();
          int2 *slots; CK(hipMalloc(&slots, (size_t)n_slots * 8)); CK(hipMemcpy(slots, hs.data(), (size_t)n_slots * 8, hipMemcpyHostToDevice));
          float *out2; CK(hipMalloc(&out2, (size_t)n_slots * 4));
          const int grid = (n_slots + 63) / 64;
          for (int var = 0; var < 4; ++var) {
            float best = 1e30f;
            for (int it = 0; it < 4; ++it) {
              CK(hipEventRecord(e0, 0));
              if (G == 4) {
                if (var == 0) fmg_kernel<4, 64, 2><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else if (var == 1) fmg_kernel<4, 64, 4><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else if (var == 2) fmg_kernel<4, 128, 2><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else fmg_kernel<4, 128, 3><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
              } else {
                if (var == 0) fmg_kernel<16, 64, 2><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else if (var == 1) fmg_kernel<16, 64, 4><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else if (var == 2) fmg_kernel<16, 128, 2><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
                else fmg_kernel<16, 128, 3><<<grid, 64>>>(W, x, bd, slots, n_slots, d, out2);
              }
              CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
              float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
            }
            printf("%3d rows/token (%7d pairs, %4.1f per feature) groups of %2d lanes (%2.0f %% filled) batch %3d floats, %d waves/SIMD: %.3f ms\n", rpts[ri],
                   n_pairs, (double)n_pairs / N, G, 100.0 * n_pairs / n_slots, var < 2 ? 64 : 128, var == 0 ? 2 : (var == 1 ? 4 : (var == 2 ? 2 : 3)), best);
          }
          CK(hipFree(slots)); CK(hipFree(out2));
        }
      }
      CK(hipFree(pairs)); CK(hipFree(out));
    }
  }
  return 0;
}
