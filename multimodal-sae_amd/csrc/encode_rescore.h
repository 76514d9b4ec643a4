// encode_rescore.h -- behind the candidate GEMM: per-token candidate select + exact f32 re-score + verification
// (select_rescore_kernel), the feature-sharded group's record packing (pack_candidates_kernel), and their launchers.
// Host dispatch: encode_fused.hip.
#pragma once
#include <type_traits>

#include "encode_defs.h"

namespace {

// ---- candidate select + exact re-score ----------------------------------------------------------
static_assert(MSAE_RESCORE_U * 4 == 64, "one re-scoring batch must be the 64 floats fast_shape_ok() guarantees");
struct RescoreArgs {
  const float *a32; const float *W_enc, *b_enc;
  const float *tau_vals; int tau_ld, tau_col;
  const int *cnt; const unsigned long long *cand; int cap;
  int T, d, N, k, r_max;
  int set_feature; float set_value; int zero_feature;
  const f32x4 *rowc, *colc;           // error-band constants per token / per feature
  float zz12, z2; int i8;
  float zc2;                          // (model check) a re-scored pair further than sqrt(zc2) sigma from its coarse value flags the token
  float *vals; int32_t *idx; int64_t *idx64; int32_t *status;   // idx / idx64: either may be null
  int *flagged; int *n_flagged; int fb_cap;
  int32_t *rows_out;                  // optional diagnostics (msae_options::rows_rescored)
  // EXT (feature-sharded group, msae_rescore_candidates): the candidate lists come as the shards' records
  // instead of cnt / cand / tau_vals / rowc / colc: record (g, t) at ext + ((size_t)g * ext_T + t) * ext_stride
  const unsigned char *ext; int ext_G, ext_C, ext_T, ext_stride, ext_valid;
  int lpr;   // lanes per row in the first round (1, 2, 4): small batches need the extra bytes in flight (rescore_shape)
  // FEATURE-MAJOR first round (PHASE 1 / 2 of select_rescore_kernel, fm_* kernels below): per-feature pair counts [N + 1],
  // first-round size per token (| FM_SORTED), the first-round keys [T][fm_rcap], their exact pre-activations [T][fm_rcap]
  int *fm_count; int *fm_target; unsigned long long *fm_keys; const float *fm_pre; int fm_rcap;
  unsigned long long *fm_cand;        // = cand, writable: PHASE 1 leaves a fully sorted list there for PHASE 2
  int *fm_rank;                       // = fm_pre's storage: a pair's rank inside its feature, between PHASE 1 and the scatter
  // LEAN / full pairs of PHASE kernels: fm_defer[(PHASE - 1) * T + t] != 0 <=> the LEAN launch left token t to the full one;
  // fm_all != 0: this (full) launch takes every token (no LEAN launch ran for the phase)
  int *fm_defer; int fm_all;
};
// fm_target[t] = first-round size (12 bits) | sorted prefix saved in fm_keys (8 bits, PHASE 1's preselect) << 12 | FM_SORTED
constexpr int FM_SORTED = 1 << 30;   // the token's whole list was written back to cand in sorted order
constexpr int FM_TARGET_MASK = 0xFFF, FM_PREFIX_SHIFT = 12, FM_PREFIX_MASK = 0xFF;

// One shard's record of a token (msae_shard_candidates): C keys (order key of the upper value u | 0x7FFFFFFF -
// GLOBAL feature, 0 = empty), C times z sigma of that (token, feature) pair, tau = the largest u any feature of
// the shard NOT in the record can have (+inf: the shard could not bound it -> the token is recomputed exactly).
__host__ __device__ inline int shard_record_bytes(int C) { return C * 12 + 8; }

// Wave-wide bitonic sort (descending) of n = power-of-two u64 keys in LDS by ONE 64-lane wave.
template <int NT>
__device__ __forceinline__ void wave_sort_desc_u64(unsigned long long *s, int n, int lane) {
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = lane; i < (n >> 1); i += NT) {
        const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = s[lo], y = s[hi];
        if ((x < y) == desc) { s[lo] = y; s[hi] = x; }
      }
    }
  __syncthreads();
}

// The same order for <= 64 R keys by ONE wave in registers: key i = r * 64 + lane sits in v[r]; partners 64 or more apart are
// the lane's own registers, closer ones another lane's (two 32-bit shuffles).  No LDS traffic, no barriers: a 64-key sort is
// 21 compare-exchange steps of ~10 instructions (the LDS version: 21 barriers, ~10 k cycles for a wave that is alone).
template <int R>
__device__ __forceinline__ void wave_sort_desc_u64_regs(unsigned long long (&v)[R], int lane) {
#pragma unroll
  for (int size = 2; size <= 64 * R; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64) {
        constexpr int dummy = 0; (void)dummy;
        const int rs = stride >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & rs) == 0) {
            const bool desc = (((r * 64 + lane) & size) == 0);
            const unsigned long long a = v[r], b = v[r | rs];
            if ((a < b) == desc) { v[r] = b; v[r | rs] = a; }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = r * 64 + lane;
          const unsigned lo = __shfl_xor((unsigned)v[r], stride, 64), hi = __shfl_xor((unsigned)(v[r] >> 32), stride, 64);
          const unsigned long long o = ((unsigned long long)hi << 32) | lo;
          // the lower index of a pair keeps the larger key where the block sorts descending
          const bool lower = (lane & stride) == 0, desc = ((i & size) == 0);
          const bool take_max = lower == desc;
          v[r] = take_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
        }
      }
    }
}

// ... and for 64 NW R keys by a WORKGROUP of NW waves: key i = tid * R + r sits in v[r] of thread tid.  Partners closer than R
// are the thread's own registers, up to 32 R apart another lane's (shuffles), farther another wave's: those few steps
// (3 of 66 for 2048 keys) go through `xch` (LDS, 64 NW R keys) behind barriers.  All threads call.
template <int NW, int R>
__device__ __forceinline__ void wg_sort_desc_u64_regs(unsigned long long (&v)[R], int tid, unsigned long long *xch) {
  constexpr int M = 64 * NW * R;
#pragma unroll
  for (int size = 2; size <= M; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride < R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & stride) == 0) {
            const bool desc = (((tid * R + r) & size) == 0);
            const unsigned long long a = v[r], b = v[r | stride];
            if ((a < b) == desc) { v[r] = b; v[r | stride] = a; }
          }
        }
      } else {
        const int pt = stride / R;                         // partner thread = tid ^ pt
        const bool lower = (tid & pt) == 0;
        if (pt >= 64) {
          __syncthreads();
#pragma unroll
          for (int r = 0; r < R; ++r) xch[tid * R + r] = v[r];
          __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          unsigned long long o;
          if (pt >= 64) o = xch[(tid ^ pt) * R + r];
          else o = ((unsigned long long)__shfl_xor((unsigned)(v[r] >> 32), pt, 64) << 32) | __shfl_xor((unsigned)v[r], pt, 64);
          const bool desc = (((tid * R + r) & size) == 0);
          const bool take_max = lower == desc;
          v[r] = take_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
        }
      }
    }
}

// number of keys (sorted descending, value in the upper 32 bits as an order key) whose value is >= v
__device__ __forceinline__ int count_ge(const unsigned long long *keys, int n, float v) {
  const unsigned tk = f32_order_key(v);
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((unsigned)(keys[mid] >> 32) >= tk) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ONE WAVE per token (64-thread workgroup; 4 waves for k > 64).  dynamic LDS: keys[cap] u64 | res[nrp] u64.
//
// The candidate list is ordered by the UPPER value u = coarse + z*sigma; lane c re-scores candidate c
// with the exact ascending-k f32 chain: it walks row f of W_enc with two software-pipelined batches
// of 16 x 16-B loads (256 B = two lines per batch) while the token's f32 activation vector a32[t][:]
// arrives through wave-uniform scalar loads.  No LDS staging of operands: the data in flight lives in
// VGPRs (7 waves x ~45 lanes x 512 B per CU), which is what keeps the HBM pipe full -- streaming the
// rows through LDS instead caps it at the ring size and measured 2.5 ms vs 1.4.
// HBM-bound: ~42 rows x d x 4 B per token.
//
// Rounds.  Needed are exactly the candidates with u >= v_k (the exact k-th value): everything else
// has p <= u < v_k.  v_k is not known beforehand, so round 1 takes the candidates with
//     u >= (k-th largest coarse value among the first NT) - zeta * (their median sigma)
// (the lanes look up the band of "their" candidate to get coarse = u - z*sigma), which is the needed
// set plus about one row in 96 % of the tokens; the exact v_k of round 1 is a lower bound of the final
// one, so ONE extension to every u >= v_k completes the rest.  A token verifies when
//     all candidates with u >= v_k are re-scored  and  v_k > tau  (non-candidates have u <= tau)
// and no re-scored pair contradicted the error model (|p - coarse| <= 6 sigma).  Tokens that fail (or
// overflowed their list / have tau <= 0 / more than r_max rows to read) go to the exact path.
// EXT: the list is the union of the shards' records; keys[] then carries the list POSITION in its low word
// (feature and z sigma are looked up by position: ef[], ezs[]).
// LDSA (small batches, p.lpr > 1): the token's activations are copied to LDS once and every lane reads the 16 B that
// belong to ITS piece of the row (ds_read_b128, counted waits) -- the scalar loads of the default path return out of
// order, so each pair of them is a full lgkmcnt(0) round trip (128 per pass), which nothing hides when a token's
// waves are alone on their SIMDs.
//
// PHASE (feature-major first round, run_fast): when a feature is a candidate of many tokens of the batch (k = 256 at 8192 tokens:
// 347 rows per token = 22 tokens per feature), the token-major first round reads every row of W_enc ~22 times from HBM.  PHASE 1
// stops behind the choice of the first round and hands its (token, feature) pairs to a counting sort by feature; fm_dot_kernel
// computes the same exact chains feature-major (W_enc once, the activations out of the Infinity Cache); PHASE 2 picks the
// values up as its first round and continues as PHASE 0 does (verification, follow-up rounds token-major: a handful of rows).
// LEAN (PHASE 1 / 2): the PHASE kernels are chains of dependent steps of one lone workgroup per token, i.e. latency; how many
// tokens a CU works on at a time is set by the LDS a workgroup claims, and that is sized for the worst token (the whole list:
// 16-32 KB) although nearly every token is done with a sorted prefix of 128-512 keys.  A LEAN launch claims the prefix only
// (6 KB at k = 32: 28 tokens per CU instead of 4-7) and LEAVES a token to the full-size launch behind it -- before any side
// effect -- the moment it would need the whole list (a full sort, a candidate behind the prefix) or a follow-up round.
template <int NW, bool EXT = false, bool LDSA = false, int PHASE = 0, bool LEAN = false>   // NW waves per token: 1 for k <= 64, 4 for larger k (longer lists)
__global__ __launch_bounds__(64 * NW) void select_rescore_kernel(RescoreArgs p, const float *__restrict__ a32,
                                                            const float *__restrict__ W_enc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  const int nrp = next_pow2(p.r_max + 1);
  static_assert(!LEAN || (PHASE != 0 && !EXT && !LDSA), "LEAN: the PHASE kernels of the single-GPU path");
  // key slots of a LEAN launch (the 4-wave PHASE 1 sorts whole lists of up to 2048 keys in registers and needs the slots as
  // its exchange buffer only; PHASE 1 has no results: no res[] behind the keys)
  constexpr int LEAN_KEYS = NW == 1 ? 256 : (PHASE == 1 ? 2048 : 512);
  const int kcap = LEAN ? LEAN_KEYS : p.cap;
  unsigned long long *res = keys + kcap;
  [[maybe_unused]] float *ezs = reinterpret_cast<float *>(res + nrp);     // EXT only: [cap] z sigma by list position
  [[maybe_unused]] int *ef = reinterpret_cast<int *>(ezs + p.cap);        // EXT only: [cap] global feature by position
  [[maybe_unused]] float *a_lds = EXT ? reinterpret_cast<float *>(ef + p.cap) : ezs;   // LDSA only: [d]
  constexpr int NT = 64 * NW;
  __shared__ float s_cc[NT], s_zs[NT], s_pick[2];
  __shared__ int s_n;
  __shared__ unsigned s_tau;
  const int lane = threadIdx.x;   // thread index within the token's workgroup
  const int t = blockIdx.x;
  if constexpr (EXT) { if (t >= p.ext_valid) return; }
  if constexpr (PHASE != 0 && !LEAN) { if (!p.fm_all && p.fm_defer[(size_t)(PHASE - 1) * p.T + t] == 0) return; }   // the LEAN launch did it
  [[maybe_unused]] bool deferred = false;             // LEAN: this token needs the full-size launch (wave-uniform)
#define MSAE_LEAN_BAIL()                                                                   \
  do {                                                                                     \
    if constexpr (LEAN) {                                                                  \
      if (deferred) {                                                                      \
        if (lane == 0) p.fm_defer[(size_t)(PHASE - 1) * p.T + t] = 1;                      \
        return;                                                                            \
      }                                                                                    \
    }                                                                                      \
  } while (0)
  int cnt, n;
  float tau;
  MSAE_RTL(0);
  const float *__restrict__ a = a32 + (size_t)t * p.d;  // noalias kernel arg + uniform address: s_load
  auto stage_a = [&]() {
    for (int i = 4 * (int)threadIdx.x; i < p.d; i += 4 * 64 * NW)
      *reinterpret_cast<f32x4 *>(a_lds + i) = *reinterpret_cast<const f32x4 *>(a + i);
  };
  // (PHASE 2 stages them when a token gets a follow-up round: its first round read no row here)
  if constexpr (LDSA && PHASE == 0) stage_a();            // published by the barriers of the list sort below
  f32x4 rc = {0.f, 0.f, 0.f, 0.f};
  const bool i8 = p.i8 != 0;
  int np;
  if constexpr (EXT) {
    const int M = p.ext_G * p.ext_C;
    np = next_pow2(M > 2 ? M : 2);
    if (lane == 0) { s_n = 0; s_tau = 0u; }
    __syncthreads();
    int mine = 0;
    for (int i = lane; i < np; i += NT) {
      unsigned long long kv = 0ull;
      if (i < M) {
        const int g = i / p.ext_C, j = i - g * p.ext_C;
        const unsigned char *rec = p.ext + ((size_t)g * p.ext_T + t) * p.ext_stride;
        const unsigned long long key = reinterpret_cast<const unsigned long long *>(rec)[j];
        if (key != 0ull) {
          kv = (key & 0xFFFFFFFF00000000ull) | (unsigned)(0x7FFFFFFF - i);
          ef[i] = rank_key_index(key);
          ezs[i] = reinterpret_cast<const float *>(rec + (size_t)p.ext_C * 8)[j];
          ++mine;
        }
      }
      keys[i] = kv;
    }
    if (mine) atomicAdd(&s_n, mine);
    for (int g = lane; g < p.ext_G; g += NT) {   // tau = the largest bound of ALL shards (order keys: +inf dominates, NaN never enters)
      const unsigned char *rec = p.ext + ((size_t)g * p.ext_T + t) * p.ext_stride;
      atomicMax(&s_tau, f32_order_key(*reinterpret_cast<const float *>(rec + (size_t)p.ext_C * 12)));
    }
    __syncthreads();
    n = cnt = s_n;
    tau = f32_from_order_key(s_tau);
  } else {
    cnt = p.cnt[t];
    n = cnt < p.cap ? cnt : p.cap;
    tau = p.tau_vals[(size_t)t * p.tau_ld + p.tau_col];
    rc = p.rowc[t];
    np = next_pow2(n > 2 ? n : 2);
  }
  if constexpr (PHASE != 1) { for (int i = lane; i < nrp; i += NT) res[i] = 0ull; }
  MSAE_RTL(1);
  // keys[0, n_sorted) hold the n_sorted largest keys in descending order (upper value desc, index asc on ties).
  // PARTIAL: of a list of ~650 candidates a token uses the first 40-60, so one wave first SELECTS its PRE_LO..PRE_HI
  // largest (keys in registers, bisection on the value word with ballot counts, a handful of steps) and sorts only
  // those 128 slots; whoever then needs a candidate behind them (count_needed, the target check of a round) gets the
  // full sort after all -- the presorted prefix is the same keys in the same places.
  constexpr int PRE_LO = 96, PRE_HI = 128, PRE_MIN = 192, PRE_PK = 32;
  int n_sorted = n;
  bool partial = false;
  bool presorted = false;                      // PHASE 2: PHASE 1 left the list sorted in place
  if constexpr (PHASE == 2) presorted = (p.fm_target[t] & FM_SORTED) != 0;
  bool have_keys = false;                      // keys[0, n_sorted) already hold a sorted prefix of the list
  auto full_sort = [&]() {
    if constexpr (LEAN) { if (np > LEAN_KEYS) { deferred = true; return; } }
    if constexpr (!EXT) {
      __syncthreads();
      for (int i = lane; i < np; i += NT) keys[i] = (i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
    }
    if (presorted) { __syncthreads(); return; }
    if constexpr (NW == 4 && !EXT) {
      if (np <= 1024 && kcap >= 1024) {                    // (round 6) half of the k = 256 tokens have <= 1024 candidates: 4 keys per thread, 55 stages instead of 66
        __syncthreads();
        unsigned long long v4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = lane * 4 + r < np ? keys[lane * 4 + r] : 0ull;
        wg_sort_desc_u64_regs<4, 4>(v4, lane, keys);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) keys[lane * 4 + r] = v4[r];
        __syncthreads();
        return;
      }
      if (np <= 2048 && kcap >= 2048) {                    // wave-uniform: 8 keys per thread, sorted in registers (keys[] = the exchange buffer: 2048 slots)
        __syncthreads();
        unsigned long long v8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = lane * 8 + r < np ? keys[lane * 8 + r] : 0ull;
        wg_sort_desc_u64_regs<4, 8>(v8, lane, keys);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) keys[lane * 8 + r] = v8[r];
        __syncthreads();
        return;
      }
    }
    wave_sort_desc_u64<NT>(keys, np, lane);
  };
  if constexpr (PHASE == 2 && NW == 1) {      // PHASE 1's preselected + sorted prefix, as it left it
    const int saved = (p.fm_target[t] >> FM_PREFIX_SHIFT) & FM_PREFIX_MASK;
    if (!presorted && saved > 0) {
      for (int i = lane; i < PRE_HI; i += NT) keys[i] = i < saved ? p.fm_keys[(size_t)t * p.fm_rcap + i] : 0ull;
      __syncthreads();
      partial = true;
      n_sorted = saved;
      have_keys = true;
    }
  }
  if constexpr (PHASE == 2 && LEAN) {          // ... or the first LEAN_KEYS of the list PHASE 1 sorted in place
    if (presorted && n > LEAN_KEYS) {
      for (int i = lane; i < LEAN_KEYS; i += NT) keys[i] = p.cand[(size_t)t * p.cap + i];
      __syncthreads();
      partial = true;
      n_sorted = LEAN_KEYS;
      have_keys = true;
    }
  }
  if constexpr (!EXT && NW == 1) {
    if (msae_tuning::RESCORE_PRESELECT && !presorted && !partial && n > PRE_MIN && n <= 64 * PRE_PK && p.k + 4 <= 64) {          // wave-uniform
      const int nj = (n + 63) >> 6;
      unsigned long long kreg[PRE_PK];
#pragma unroll
      for (int j = 0; j < PRE_PK; ++j) {
        const int i = j * 64 + lane;
        kreg[j] = (j < nj && i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
      }
      unsigned lo = 0u, hi = 0xFFFFFFFFu;     // count(value word >= lo) > PRE_HI, count(>= hi) < PRE_LO
      int c_sel = -1;
      unsigned thr = 0u;
      while (hi - lo > 1u) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int jb = 0; jb < PRE_PK; jb += 8) {             // one branch per eight key slots (empty slots hold 0)
          if (jb < nj) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j)
              c += __builtin_popcountll(__builtin_amdgcn_ballot_w64((unsigned)(kreg[j] >> 32) >= mid));
          }
        }
        if (c > PRE_HI) lo = mid;
        else if (c < PRE_LO) hi = mid;
        else { c_sel = c; thr = mid; break; }
      }
      if (c_sel > 0) {                                       // (ties across the window: no such threshold -> full sort)
        int base = 0;
#pragma unroll
        for (int j = 0; j < PRE_PK; ++j) {
          if (j < nj) {
            const bool take = (unsigned)(kreg[j] >> 32) >= thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
            if (take) keys[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = kreg[j];
            base += __builtin_popcountll(m);
          }
        }
        for (int i = c_sel + lane; i < PRE_HI; i += NT) keys[i] = 0ull;
        {   // the PRE_HI = 128 slots: two keys per lane, sorted in registers
          static_assert(PRE_HI == 2 * NT, "two key slots per lane");
          __syncthreads();
          unsigned long long v2[2] = {keys[lane], keys[64 + lane]};
          wave_sort_desc_u64_regs<2>(v2, lane);
          keys[lane] = v2[0]; keys[64 + lane] = v2[1];
          __syncthreads();
        }
        partial = true;
        n_sorted = c_sel;
        have_keys = true;
      }
    }
  }
  if (!have_keys) full_sort();
  MSAE_LEAN_BAIL();
  auto need_full = [&]() {
    if constexpr (LEAN) { deferred = true; return; }
    full_sort(); partial = false; n_sorted = n;
  };
  auto count_needed = [&](float v) {         // candidates with u >= v (over the whole list)
    int c = count_ge(keys, n_sorted, v);
    if (partial && c >= n_sorted) { need_full(); if (!deferred) c = count_ge(keys, n, v); }
    return c;
  };
  MSAE_RTL(2);
  const int has_set = p.set_feature >= 0 ? 1 : 0;
  if constexpr (PHASE != 1) { if (lane == 0 && has_set) res[0] = rank_key(p.set_value, p.set_feature); }

  // ---- size of the first round ------------------------------------------------------------------
  const int lim = n < p.r_max ? n : p.r_max;
  int target = lim;
  if constexpr (PHASE == 2) {
    target = p.fm_target[t] & FM_TARGET_MASK;
  } else {
    const int mt_max = p.k <= 64 ? 64 : NT;       // the same statistic whatever the number of waves per token
    const int mt = n < mt_max ? n : mt_max;
    float my_cc = -__builtin_inff(), my_zs = 0.f;
    if (lane < mt) {
      const unsigned long long key = keys[lane];
      if constexpr (EXT) my_zs = ezs[rank_key_index(key)];
      else my_zs = __builtin_sqrtf(band_sq(rc, p.colc[rank_key_index(key)], p.zz12, i8));
      my_cc = f32_from_order_key((unsigned)(key >> 32)) - my_zs;
    }
    s_cc[lane] = my_cc;
    s_zs[lane] = my_zs;
    if (lane < 2) s_pick[lane] = lane == 0 ? -__builtin_inff() : 0.f;
    __syncthreads();
    const int kk = p.k - has_set;
    if (lane < mt && kk >= 1 && kk <= mt) {
      int rank_c = 0, rank_z = 0;
      for (int j = 0; j < mt; ++j) {
        const float cj = s_cc[j], zj = s_zs[j];
        rank_c += (cj > my_cc || (cj == my_cc && j < lane)) ? 1 : 0;
        rank_z += (zj < my_zs || (zj == my_zs && j < lane)) ? 1 : 0;
      }
      if (rank_c == kk - 1) s_pick[0] = my_cc;
      if (rank_z == mt / 2) s_pick[1] = my_zs;
    }
    __syncthreads();
    if (kk >= 1 && kk <= mt && p.z2 > 0.f) {
      const float thr1 = s_pick[0] - GUARD_ZETA * s_pick[1] * __builtin_amdgcn_rsqf(p.z2);
      int n1 = count_needed(thr1);
      MSAE_LEAN_BAIL();
      if (n1 < p.k + 4) n1 = p.k + 4;
      target = n1 < lim ? n1 : lim;
    }
  }
  if constexpr (LDSA && PHASE == 0) {   // small batch: one pass reads 64 NW / lpr rows whatever the target -- fill it (fewer second rounds)
    const int rpp = p.lpr > 0 ? NT / p.lpr : NT;
    const int fill = rpp < lim ? rpp : lim;
    if (target < fill) target = fill;
  }

  MSAE_RTL(3);
  const float zc2 = p.zc2;
  const bool guarded = !EXT && rc[3] != 0.f;     // the token's shape is outside the noise model (quant_x_kernel): exact path
  if (guarded) target = 0;                       // (no row is read for it here)
  if constexpr (PHASE == 1) {
    if (partial && target > n_sorted) need_full();          // wave-uniform
    MSAE_LEAN_BAIL();
    // the first round's keys (the counting sort reads them), and behind them the rest of a preselected prefix for PHASE 2
    const int save = partial && n_sorted <= p.fm_rcap && n_sorted <= FM_PREFIX_MASK ? n_sorted : 0;
    for (int c = lane; c < (target > save ? target : save); c += NT) {
      const unsigned long long key = keys[c];
      p.fm_keys[(size_t)t * p.fm_rcap + c] = key;
      // the count's old value is this pair's rank among its feature's pairs: the scatter needs no second atomic (the rank
      // waits in the pair's slot of fm_pre, which fm_dot_kernel fills later)
      if (c < target) p.fm_rank[(size_t)t * p.fm_rcap + c] = atomicAdd(p.fm_count + rank_key_index(key), 1);
    }
    if (!partial)
      for (int i = lane; i < n; i += NT) p.fm_cand[(size_t)t * p.cap + i] = keys[i];
    if (lane == 0) p.fm_target[t] = target | (save << FM_PREFIX_SHIFT) | (partial ? 0 : FM_SORTED);
    return;
  }
  int done = 0;                                  // candidates re-scored so far (wave-uniform)
  bool ok = false, viol = false;
  int rounds = 0;
  const int first_target = target;
  for (;;) {
    ++rounds;
    int my_viol = 0;
    // LPR = 1: lane c streams row c (16 B per lane and instruction).  LPR = 4 (tuning builds): four lanes share a
    // row, lane q loading bytes [16 q, 16 q + 16) of every 64-B piece -- four times fewer cache lines per
    // instruction, but only 16 rows per pass, i.e. three row-streaming latencies per round instead of one.  The
    // chain stays one serial ascending-k sequence: sub-step q multiplies the group's lane-q piece (every lane
    // executes it on its own registers; only lane q's is the true partial sum) and a quad rotate hands the
    // accumulator on.  The activations are wave-uniform scalar operands either way.
    auto run_pass = [&](auto lpr_tag) {
      constexpr int LPR = decltype(lpr_tag)::value;
      constexpr int RPP = NT / LPR;                  // rows per pass
      constexpr int RS_U = MSAE_RESCORE_U, RS_B = 4 * RS_U * LPR;   // floats of a row per batch
      const int rq = lane / LPR, q = lane % LPR;
      for (int c0 = done; c0 < target; c0 += RPP) {
        const int c = c0 + rq;
        const bool active = c < target;
        const unsigned long long key = active ? keys[c] : keys[c0];
        int f = rank_key_index(key);
        float ext_zs = 0.f;
        f32x4 cc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EXT) { ext_zs = ezs[f]; f = ef[f]; }     // list position -> (z sigma, global feature)
        else cc = p.colc[f];
        const float upper = f32_from_order_key((unsigned)(key >> 32));
        const float *__restrict__ w = W_enc + (size_t)f * p.d + 4 * q;
        float acc = 0.f;
        // two batches of RS_U x 16 B per lane, software-pipelined: while one batch is consumed the
        // other is in flight, so the lane never drains its loads
        f32x4 wa[RS_U], wb[RS_U];
        auto fetch = [&](f32x4 (&dst)[RS_U], int kk) {
#pragma unroll
          for (int u = 0; u < RS_U; ++u) dst[u] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * LPR * u);
        };
        auto consume = [&](const f32x4 (&src)[RS_U], int kk) {
#pragma unroll
          for (int u = 0; u < RS_U; ++u) {
            [[maybe_unused]] f32x4 av;                   // LDSA: the activations of this lane's own piece
            if constexpr (LDSA) av = *reinterpret_cast<const f32x4 *>(a_lds + kk + 4 * LPR * u + 4 * q);
#pragma unroll
            for (int qq = 0; qq < LPR; ++qq) {
              const int k0 = kk + 4 * LPR * u + 4 * qq;
              if constexpr (LDSA) {                      // only sub-step qq == q carries the true partial sum
                acc = __builtin_fmaf(av[0], src[u][0], acc);
                acc = __builtin_fmaf(av[1], src[u][1], acc);
                acc = __builtin_fmaf(av[2], src[u][2], acc);
                acc = __builtin_fmaf(av[3], src[u][3], acc);
              } else {
                acc = __builtin_fmaf(a[k0 + 0], src[u][0], acc);   // a[] is wave-uniform: SGPRs
                acc = __builtin_fmaf(a[k0 + 1], src[u][1], acc);
                acc = __builtin_fmaf(a[k0 + 2], src[u][2], acc);
                acc = __builtin_fmaf(a[k0 + 3], src[u][3], acc);
              }
              if constexpr (LPR == 4)   // quad_perm:[3,0,1,2] -- lane i takes lane i - 1's value, lane 0 lane 3's
                acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x93, 0xF, 0xF, false));
              if constexpr (LPR == 2)   // quad_perm:[1,0,3,2] -- the two lanes of a pair swap
                acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, false));
            }
          }
        };
        fetch(wa, 0);
        for (int kk = 0; kk < p.d; kk += 2 * RS_B) {     // d % RS_B == 0 (fast_shape_ok / the caller's choice of LPR)
          const bool has_b = kk + RS_B < p.d;
          if (has_b) fetch(wb, kk + RS_B);
          consume(wa, kk);
          if (kk + 2 * RS_B < p.d) fetch(wa, kk + 2 * RS_B);
          if (has_b) consume(wb, kk + RS_B);
        }
        const float pre = acc + (p.b_enc ? p.b_enc[f] : 0.f);
        if (active && q == 0) {                          // whole pieces done: the sum is back in the group's lane 0
          res[has_set + c] = rank_key(pre > 0.f ? pre : 0.f, f);  // slots past the sorted prefix are 0
          // model check: |p - coarse| <= 6 sigma  <=>  (p - coarse)^2 z^2 <= 36 (z sigma)^2
          const float zs2 = EXT ? ext_zs * ext_zs : band_sq(rc, cc, p.zz12, i8);
          const float diff = pre - (upper - __builtin_sqrtf(zs2));
          if (diff * diff * p.z2 > zc2 * zs2 * 1.0001f + 1e-30f) my_viol = 1;
        }
      }
    };
    // A follow-up round re-scores a handful of rows: with a lane per row each of them is a latency chain (16 KB at
    // 512 B in flight = 32 round trips, ~60 us whatever the load); four lanes per row carry 2 KB in flight each.
    // The first round of a SMALL batch (too few tokens to fill the chip with a lane per row) does the same with
    // p.lpr lanes per row and as many waves per token.
    if (partial && target > n_sorted) need_full();          // wave-uniform
    MSAE_LEAN_BAIL();
    bool from_fm = false;
    if constexpr (PHASE == 2) {
      if (rounds == 1) {                                     // the first round's values: fm_dot_kernel computed them
        from_fm = true;
        for (int c = lane; c < target; c += NT) {
          const unsigned long long key = keys[c];
          const int f = rank_key_index(key);
          const float upper = f32_from_order_key((unsigned)(key >> 32));
          const float pre = p.fm_pre[(size_t)t * p.fm_rcap + c];
          res[has_set + c] = rank_key(pre > 0.f ? pre : 0.f, f);
          const float zs2 = band_sq(rc, p.colc[f], p.zz12, i8);
          const float diff = pre - (upper - __builtin_sqrtf(zs2));
          if (diff * diff * p.z2 > zc2 * zs2 * 1.0001f + 1e-30f) my_viol = 1;
        }
      }
    }
    if (!from_fm) {
      // PHASE 2: a follow-up round's wave is alone on its SIMD -- the scalar loads of the activations would be one exposed round
      // trip per pair of them (~130 k cycles per pass, the kernel's tail); LDSA reads them from LDS
      if constexpr (LDSA && PHASE == 2) { if (rounds == 2) { stage_a(); __syncthreads(); } }
      const bool few = rounds > 1 && target - done <= NT / 4;
      int lpr = few ? 4 : (MSAE_RESCORE_LPR == 4 ? 4 : p.lpr);
      while (lpr > 1 && p.d % (4 * MSAE_RESCORE_U * lpr) != 0) lpr >>= 1;      // a batch is 64 lpr floats of a row
      if (lpr == 4) run_pass(std::integral_constant<int, 4>());
      else if (lpr == 2) run_pass(std::integral_constant<int, 2>());
      else run_pass(std::integral_constant<int, 1>());
    }
    done = target;
    viol = viol || (__syncthreads_or(my_viol) != 0);
    MSAE_RTL(2 + 2 * rounds);
    {   // res[] is zero (= empty, the smallest key) behind the slots written so far: sort the filled prefix only
      const int filled = next_pow2(done + has_set > 2 ? done + has_set : 2);
      bool in_regs = false;
      if constexpr (NW == 1) {
        if (filled <= 64 && nrp >= 64) {                     // wave-uniform: one key per lane, sorted in registers
          in_regs = true;
          __syncthreads();
          unsigned long long v1[1] = {res[lane]};
          wave_sort_desc_u64_regs<1>(v1, lane);
          res[lane] = v1[0];
          __syncthreads();
        }
      }
      if constexpr (NW == 4) {
        if (filled <= 1024 && nrp >= 1024) {                // wave-uniform: 4 keys per thread (k = 256: ~350 results)
          in_regs = true;
          __syncthreads();
          unsigned long long v4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v4[r] = res[lane * 4 + r];
          wg_sort_desc_u64_regs<4, 4>(v4, lane, res);
          __syncthreads();
#pragma unroll
          for (int r = 0; r < 4; ++r) res[lane * 4 + r] = v4[r];
          __syncthreads();
        }
      }
      if (!in_regs) wave_sort_desc_u64<NT>(res, filled < nrp ? filled : nrp, lane);
    }
    MSAE_RTL(3 + 2 * rounds);
    const bool have_k = done + has_set >= p.k;
    const float v_k = f32_from_order_key((unsigned)(res[p.k - 1] >> 32));
    const int needed = have_k ? count_needed(v_k) : n;          // candidates with u >= v_k
    MSAE_LEAN_BAIL();
    ok = (cnt <= p.cap) && (tau > 0.f) && have_k && !viol && needed <= done && v_k > tau * 1.000001f && !guarded;
    if (ok || viol || guarded || done >= lim || !(tau > 0.f) || cnt > p.cap) break;
    if constexpr (LEAN) { deferred = true; MSAE_LEAN_BAIL(); }   // a follow-up round: the full-size launch (rows token-major, activations in LDS)
    target = needed > done ? needed : done + 1;
    if (target > lim) target = lim;
    __syncthreads();
  }

  MSAE_RTL(14);
  MSAE_RTL_VALUE(15, ((unsigned long long)rounds << 32) | (unsigned)done);
  for (int j = lane; j < p.k; j += NT) {
    const unsigned long long key = res[j];
    const int fi = key ? rank_key_index(key) : 0;
    if (p.idx) p.idx[(size_t)t * p.k + j] = fi;
    if (p.idx64) p.idx64[(size_t)t * p.k + j] = fi;
    p.vals[(size_t)t * p.k + j] = key ? f32_from_order_key((unsigned)(key >> 32)) : 0.f;
  }
  if (lane == 0) {
    // not verified: 2 | reason bits (4 list overflow, 8 tau <= 0, 16 fewer than k candidates,
    // 32 more than r_max rows needed / v_k not above tau, 64 a re-scored pair contradicted the error
    // model); the exact fallback rewrites it to 1 once it has recomputed t
    const int reason = 2 | (cnt > p.cap ? 4 : 0) | (!(tau > 0.f) ? 8 : 0) |
                       (done + has_set < p.k ? 16 : 0) | (guarded ? 128 : (viol ? 64 : 32));
    if (p.status) p.status[t] = ok ? 0 : reason;
    // msae_options::rows_rescored: [1 << 30: first round feature-major] | rounds << 24 | first-round rows << 12 | rows of W_enc
    // this token read (0: not verified here)
    if (p.rows_out) p.rows_out[t] = ok ? (PHASE == 2 ? 1 << 30 : 0) | ((rounds & 0x3F) << 24) | ((first_target < 0xFFF ? first_target : 0xFFF) << 12) | (done < 0xFFF ? done : 0xFFF) : 0;   // (12-bit fields saturate)
    if (!ok) {
      const int slot = atomicAdd(p.n_flagged, 1);
      if (slot < p.fb_cap) p.flagged[slot] = t;
    }
  }
}

#undef MSAE_LEAN_BAIL

// ---- feature-major first round -------------------------------------------------------------------
// The pairs of a feature occupy whole GROUPS of G lanes (G = 4 or 16: fm_group_lanes) of fm_dot_kernel's waves, so a feature's
// count is rounded up to a multiple of G; the pad slots carry the feature and no token.
// counts[0 .. N) -> padded exclusive starts in place, counts[N] = the number of slots: block sums (1024 counts each), then
// every block adds up the sums before it and scans its own 1024 counts (4 per thread) and writes its features' pad slots.
constexpr int FM_SCAN_BLOCK = 1024;
__device__ __forceinline__ int fm_pad(int c, int G) { return (c + G - 1) & ~(G - 1); }
__global__ __launch_bounds__(256) void fm_blocksum_kernel(const int *__restrict__ counts, int N, int G, int *__restrict__ bsum) {
  __shared__ int red[4];
  const int i0 = blockIdx.x * FM_SCAN_BLOCK + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) s += (i0 + e < N) ? fm_pad(counts[i0 + e], G) : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void fm_scan_kernel(int *__restrict__ counts, int N, int G, const int *__restrict__ bsum,
                                                      int2 *__restrict__ slots) {
  __shared__ int part[256];
  __shared__ int s_base;
  const int tid = threadIdx.x, b = blockIdx.x;
  int pre = 0;
  for (int j = tid; j < b; j += 256) pre += bsum[j];
  part[tid] = pre;
  __syncthreads();
  if (tid < 64) {
    int v = part[tid] + part[tid + 64] + part[tid + 128] + part[tid + 192];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (tid == 0) s_base = v;
  }
  __syncthreads();
  const int i0 = b * FM_SCAN_BLOCK + tid * 4;
  int c[4], sum = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { c[e] = (i0 + e < N) ? counts[i0 + e] : 0; sum += fm_pad(c[e], G); }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = s_base + part[tid] - sum;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (i0 + e < N) {
      counts[i0 + e] = run;
      const int padded = fm_pad(c[e], G);
      for (int q = c[e]; q < padded; ++q) slots[run + q] = make_int2(i0 + e, -1);
      run += padded;
    }
  }
  if (b == (int)gridDim.x - 1 && tid == 255) counts[N] = s_base + part[255];
}
// pair (t, c) -> its place among its feature's slots: the feature's start + the rank PHASE 1's counting atomic returned (any
// order inside a feature: the pairs are independent).  slots[pos] = (feature, t * rcap + c).
__global__ __launch_bounds__(256) void fm_scatter_kernel(const int *__restrict__ fm_target, const unsigned long long *__restrict__ fm_keys,
                                                         const int *__restrict__ fm_rank, int rcap, const int *__restrict__ starts,
                                                         int2 *__restrict__ slots) {
  const int t = blockIdx.x, target = fm_target[t] & FM_TARGET_MASK;
  for (int c = threadIdx.x; c < target; c += 256) {
    const int f = rank_key_index(fm_keys[(size_t)t * rcap + c]);
    slots[starts[f] + fm_rank[(size_t)t * rcap + c]] = make_int2(f, t * rcap + c);
  }
}
// acc = fma(a, w of lane SH of this lane's group, acc): ONE v_fmac_f32 (fused) with the DPP source modifier on w.  Written as
// asm because the compiler does not fold a v_mov_dpp into the fma: it emits the 64 moves of a batch up front (+128 VGPRs).
// (The registers read through DPP are written by loads, not by VALU instructions; the s_nop covers a copy the register
// allocator might put in front of a chain.)
template <int G, int SH>
__device__ __forceinline__ void fma_share(float &acc, float a, float w) {
  if constexpr (G == 16) {
    if constexpr (SH == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a));
    else asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a), "n"(SH));
  } else {
    if constexpr (SH == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a));
    else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(a), "n"(SH));
  }
}
// piece P of a batch: the 4 G consecutive elements the group's G lanes hold (4 each), in ascending k
template <int DT, bool HAS_BD, int G, int P, int SH = 0>
struct FmChain {
  template <class WR, class XR>
  static __device__ __forceinline__ void run(float &acc, const WR &sw, const XR &sx, const float *__restrict__ bd) {
    if constexpr (SH < G) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = (P * G + SH) * 4 + c;                // element of the batch
        float xf;
        if constexpr (DT == MSAE_F32) xf = __uint_as_float(sx[e >> 2][e & 3]);
        else {
          const unsigned pk = sx[e >> 3][(e >> 1) & 3];
          if constexpr (DT == MSAE_BF16) xf = __uint_as_float((e & 1) ? (pk & 0xFFFF0000u) : (pk << 16));
          else xf = f16_bits_to_f32((unsigned short)((e & 1) ? (pk >> 16) : (pk & 0xFFFFu)));
        }
        float a = xf;
        if constexpr (HAS_BD) a = xf - bd[e];               // wave-uniform address: scalar loads
        fma_share<G, SH>(acc, a, sw[P][c]);
      }
      FmChain<DT, HAS_BD, G, P, SH + 1>::run(acc, sw, sx, bd);
    }
  }
};
// One lane per slot, the slots in feature order, G lanes per group: every lane walks its token's row of x itself (the
// caller's x in its own type: a = float(x) - b_dec, sae.py:174, the same f32 value prep writes to a32), the G lanes of a group
// load 16 G contiguous bytes of their feature's row of W_enc per instruction and take each other's elements through DPP
// inside ONE ascending-k fma chain per lane (select_rescore_kernel's chain, bit for bit).  W_enc comes from HBM once per
// feature; the activation rows (T x d: 67 MB of bf16 at 8192 x 4096) come out of the Infinity Cache, which is the bound:
// 23 GB at 8.9 TB/s for the 2.84 M pairs of k = 256 (profiles/r04_fm_rescore_probe.txt: 3.1 ms; a lane walking both rows
// itself 3.9 -- the texture path then carries 24 KB per pair instead of 9; token-major 7.2).
template <int DT, int G>
__global__ __launch_bounds__(64) void fm_dot_kernel(const void *__restrict__ x, const float *__restrict__ b_dec,
                                                    const float *__restrict__ W_enc, const float *__restrict__ b_enc,
                                                    const int2 *__restrict__ slots, const int *__restrict__ n_slots_p, int d, int rcap,
                                                    float *__restrict__ fm_pre) {
  const int n_slots = *n_slots_p;                        // a multiple of G
  if ((int)blockIdx.x * 64 >= n_slots) return;
  const int lane = threadIdx.x, slot = blockIdx.x * 64 + lane;
  // slots behind the last one: lanes of whole groups (n_slots % G == 0); they walk the last feature's row, without a token
  const int2 fo = slot < n_slots ? slots[slot] : make_int2(slots[n_slots - 1].x, -1);
  const bool valid = fo.y >= 0;
  constexpr bool X32 = DT == MSAE_F32;
  constexpr int KB = 64, WP = KB / (4 * G), XP = X32 ? KB / 4 : KB / 8;   // floats of k per batch; 16-B pieces of W / x per lane
  const float *__restrict__ w = W_enc + (size_t)fo.x * d + 4 * (lane % G);
  const unsigned char *__restrict__ xr = static_cast<const unsigned char *>(x) + (size_t)(valid ? fo.y / rcap : 0) * d * (X32 ? 4 : 2);
  f32x4 wa[WP], wb[WP];
  u32x4 xa[XP], xb[XP];
  float acc = 0.f;
  auto fetch = [&](f32x4 (&dw)[WP], u32x4 (&dx)[XP], int kk) {
#pragma unroll
    for (int p = 0; p < WP; ++p) dw[p] = *reinterpret_cast<const f32x4 *>(w + kk + 4 * G * p);
    if (valid) {
#pragma unroll
      for (int u = 0; u < XP; ++u) dx[u] = *reinterpret_cast<const u32x4 *>(xr + (size_t)kk * (X32 ? 4 : 2) + 16 * u);
    }
  };
  auto walk = [&](auto bd_tag) {
    constexpr bool HAS_BD = decltype(bd_tag)::value;
    auto consume = [&](const f32x4 (&sw)[WP], const u32x4 (&sx)[XP], int kk) {
      const float *__restrict__ bd = HAS_BD ? b_dec + kk : nullptr;
      FmChain<DT, HAS_BD, G, 0>::run(acc, sw, sx, bd);
      if constexpr (WP >= 2) FmChain<DT, HAS_BD, G, 1>::run(acc, sw, sx, bd);
      if constexpr (WP >= 4) { FmChain<DT, HAS_BD, G, 2>::run(acc, sw, sx, bd); FmChain<DT, HAS_BD, G, 3>::run(acc, sw, sx, bd); }
    };
    fetch(wa, xa, 0);
    for (int kk = 0; kk < d; kk += 2 * KB) {             // d % KB == 0 (fast_shape_ok)
      const bool has_b = kk + KB < d;
      if (has_b) fetch(wb, xb, kk + KB);
      consume(wa, xa, kk);
      if (kk + 2 * KB < d) fetch(wa, xa, kk + 2 * KB);
      if (has_b) consume(wb, xb, kk + KB);
    }
  };
  if (b_dec) walk(std::true_type()); else walk(std::false_type());
  if (valid) fm_pre[fo.y] = acc + (b_enc ? b_enc[fo.x] : 0.f);
}
static_assert(64 / (4 * 16) == 1 && 64 / (4 * 4) == 4, "fm_dot_kernel's consume covers WP = 1, 2 and 4");

// Feature-sharded group, sender side: the C best candidates of THIS shard per token by upper value, as the
// record shard_record_bytes() describes (global feature ids).  One wave per token.
struct PackArgs {
  const int *cnt; const unsigned long long *cand; int cap;
  const float *tau_vals; int tau_ld, tau_col;
  const f32x4 *rowc, *colc; float zz12; int i8;
  int C, row_offset, stride;
  unsigned char *recs;
};
template <int PK>   // key slots per lane: the list (<= cap <= 64 PK keys) lives in registers
__global__ __launch_bounds__(64) void pack_candidates_kernel(PackArgs p) {
  // The C largest of ~512 keys are a selection, not a sort: the keys sit in registers (PK per lane) and a bisection
  // on the 64-bit key -- unique: the feature id is its low word -- finds the C-th largest with one ballot count
  // per key slot and step; the survivors are compacted with ballot prefix counts (any order: the owner sorts).
  const int t = blockIdx.x, lane = threadIdx.x;
  const int cnt = p.cnt[t];
  const int n = cnt < p.cap ? cnt : p.cap;
  const float tau = p.tau_vals[(size_t)t * p.tau_ld + p.tau_col];
  // list complete, a real threshold behind it, and a token the noise model describes (rowc[3]: quant_x_kernel's guard)
  const bool bounded = cnt <= p.cap && tau > 0.f && p.rowc[t][3] == 0.f;
  const int nj = bounded ? (n + 63) >> 6 : 0;            // key slots in use (wave-uniform)
  unsigned long long kreg[PK];
#pragma unroll
  for (int j = 0; j < PK; ++j) {
    const int i = j * 64 + lane;
    kreg[j] = (j < nj && i < n) ? p.cand[(size_t)t * p.cap + i] : 0ull;
  }
  unsigned long long lo = 0ull, hi = ~0ull;              // count(key >= lo) >= C  (or everything is taken), count(>= hi) < C
  if (n > p.C) {
    while (hi - lo > 1ull) {
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int jb = 0; jb < PK; jb += 8) {               // one branch per eight key slots (empty slots hold 0 < mid)
        if (jb < nj) {
#pragma unroll
          for (int j = jb; j < jb + 8; ++j) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kreg[j] >= mid));
        }
      }
      if (c >= p.C) lo = mid; else hi = mid;
    }
  } else {
    lo = 1ull;                                           // every (non-empty) key
  }
  unsigned char *rec = p.recs + (size_t)t * p.stride;
  unsigned long long *okeys = reinterpret_cast<unsigned long long *>(rec);
  float *ozs = reinterpret_cast<float *>(rec + (size_t)p.C * 8);
  const f32x4 rc = p.rowc[t];
  int base = 0;
  unsigned long long below = 0ull;                       // largest key NOT taken
#pragma unroll
  for (int j = 0; j < PK; ++j) {
    if (j < nj) {
      const unsigned long long key = kreg[j];
      const bool take = key >= lo && key != 0ull;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
      if (take) {
        const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        const int f = rank_key_index(key);
        okeys[pos] = (key & 0xFFFFFFFF00000000ull) | (unsigned)(0x7FFFFFFF - (f + p.row_offset));
        ozs[pos] = __builtin_sqrtf(band_sq(rc, p.colc[f], p.zz12, p.i8 != 0));
      } else {
        below = key > below ? key : below;
      }
      base += __builtin_popcountll(m);
    }
  }
  for (int jj = base + lane; jj < p.C; jj += 64) { okeys[jj] = 0ull; ozs[jj] = 0.f; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(below, off, 64);
    below = o > below ? o : below;
  }
  if (lane == 0) {
    // what the shard's other features can reach: the best candidate left behind, else the threshold every
    // non-candidate stayed below; +inf when the shard cannot tell (overflowed list, degenerate token)
    float b = __builtin_inff();
    if (bounded) b = below != 0ull ? f32_from_order_key((unsigned)(below >> 32)) : tau;
    float *tail = reinterpret_cast<float *>(rec + (size_t)p.C * 12);
    tail[0] = b;
    tail[1] = 0.f;
  }
}

// waves per token and lanes per row of the first round: k > 64 -> 4 waves (longer lists); batches that cannot fill
// 256 CUs x 8 waves with a lane per row get 2 or 4 lanes per row (and waves per token) instead
inline void rescore_shape(int T, int k, int &nw, int &lpr) {
  // k > 64: 4 waves per token, a lane per row.  k = 256 reads ~350 rows per token (profiles/r03_rescore_stats_k256.txt),
  // i.e. a second, mostly idle pass -- but 6 waves per token (one pass) measured SLOWER, 9.65 vs 8.0 ms: the kernel's
  // ~230 VGPRs allow 8 waves per CU, and workgroups of 6 waves leave two of those slots empty
  // (profiles/r03_k256_nw6.txt); the stage is HBM-bound at 5.8 TB/s either way.
  nw = k <= 64 ? 1 : 4;
  lpr = 1;
  if (k <= 64) {
    const long lanes = (long)T * (k + 13);
    // Round 6 (profiles/r06_ab_rescore_lpr.txt): a workgroup of lpr waves per token is alone on its CU up to 256 tokens; one token
    // more and the kernel ends with the CU that holds TWO four-wave workgroups (0.078 -> 0.135 ms from 256 to 257 tokens) -- two
    // waves per token then beat four up to ~640 tokens (257: 0.099, 512: 0.123 against 0.152), a lane per row beyond (768: 0.157
    // against 0.235 / 0.207, 1024: 0.189 against 0.297 / 0.230).  The old rule asked only whether the lanes fill the chip.
    if (T <= 256 && lanes * 4 <= 131072) lpr = 4;
    else if (T <= 640 && lanes * 2 <= 131072) lpr = 2;
    if (const char *e = getenv("MSAE_LPR")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) lpr = v; }   // (A/B runs)
    nw = lpr;
  }
}
// bytes of the results array behind the keys in select_rescore_kernel's dynamic LDS (smem = (cap + nrp) * 8 + 64)
inline size_t full_minus_keys(size_t smem, int cap) { return smem - 64 - (size_t)cap * 8; }
template <bool EXT, int PHASE = 0>
inline int launch_select_rescore(RescoreArgs &ra, int T, int k, size_t smem, const float *a32, const float *W_enc,
                                 hipStream_t s) {
  int nw;
  rescore_shape(T, k, nw, ra.lpr);
  const bool ldsa = PHASE == 0 && ra.lpr > 1 && smem + (size_t)ra.d * 4 <= 96 * 1024;      // small batch: activations in LDS
  if (ldsa) smem += (size_t)ra.d * 4;
#define MSAE_RS_LAUNCH(NWV, LDSAV, PH)                                                                                   \
  do {                                                                                                                   \
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)select_rescore_kernel<NWV, EXT, LDSAV, PH>,                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                            \
    hipLaunchKernelGGL((select_rescore_kernel<NWV, EXT, LDSAV, PH>), dim3(T), dim3(64 * NWV), smem, s, ra, a32, W_enc);  \
  } while (0)
#define MSAE_RS_LAUNCH_L(NWV, PH)                                                                                       \
  do {                                                                                                                   \
    MSAE_HIP_TRY(hipFuncSetAttribute((const void *)select_rescore_kernel<NWV, EXT, false, PH, true>,                     \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                            \
    hipLaunchKernelGGL((select_rescore_kernel<NWV, EXT, false, PH, true>), dim3(T), dim3(64 * NWV), smem, s, ra, a32, W_enc); \
  } while (0)
  if constexpr (PHASE != 0) {                 // feature-major first round: large batches only (fm_shape_ok: a lane per row)
    // a LEAN launch (prefix-sized LDS) in front of the full-size one, which then takes the tokens the LEAN one left to it
    const size_t nres = PHASE == 1 ? 0 : full_minus_keys(smem, ra.cap);           // PHASE 1 has no results
    const size_t full = smem, lean = (size_t)(nw == 1 ? 256 : (PHASE == 1 ? 2048 : 512)) * 8 + nres + 64;
    const bool has_lean = ra.fm_defer != nullptr && lean < full && (nw == 1 || PHASE == 2 || ra.cap >= 2048);
    ra.fm_all = has_lean ? 0 : 1;
    if (has_lean) {
      smem = lean;
      if (nw == 1) MSAE_RS_LAUNCH_L(1, PHASE); else MSAE_RS_LAUNCH_L(4, PHASE);
      smem = full;
    }
    const bool lds2 = PHASE == 2 && nw == 1 && smem + (size_t)ra.d * 4 <= 64 * 1024;   // follow-up rounds: activations in LDS
    if (lds2) { smem += (size_t)ra.d * 4; MSAE_RS_LAUNCH(1, true, PHASE); }
    else if (nw == 1) MSAE_RS_LAUNCH(1, false, PHASE);
    else MSAE_RS_LAUNCH(4, false, PHASE);
  } else {
    if (ldsa) { if (nw == 2) MSAE_RS_LAUNCH(2, true, 0); else MSAE_RS_LAUNCH(4, true, 0); }
    else if (nw == 1) MSAE_RS_LAUNCH(1, false, 0);
    else if (nw == 2) MSAE_RS_LAUNCH(2, false, 0);
    else MSAE_RS_LAUNCH(4, false, 0);
  }
#undef MSAE_RS_LAUNCH
#undef MSAE_RS_LAUNCH_L
  return 0;
}

// The feature-major first round pays when a row of W_enc is a candidate of several tokens of the batch -- m = ~1.36 k T / N tokens
// per feature -- and because the activation rows it reads per pair instead come out of the Infinity Cache in the caller's own
// type.  Cost model per (token, feature) pair, from profiles/r04_fm_rescore_probe.txt: token-major 4 d bytes of HBM at 6.2 TB/s;
// feature-major (esize + 4 / m) d bytes over the fabric at 8 TB/s plus ~0.5 ns of counting sort, second kernel and a second
// select kernel per pair.  Measured at 8192 x 4096 x 131072, bf16 x (profiles/r04_fm_rescore.txt): k = 256 (m = 22) 7.94 -> 4.55
// ms, k = 32 (m = 2.7) 1.03 -> 0.82; small d (768: the pairs are cheap either way) and f32 activations at moderate m stay
// token-major.  MSAE_FM=0 / 1 forces the route (where the shape allows it).
inline bool fm_pays(int T, int k, int N, int d, int esize) {
  const char *e = getenv("MSAE_FM");          // (read at every call: a test forces the route on and off in one process)
  const int force = e ? atoi(e) : -1;
  if (force >= 0) return force != 0;
  const double m = 1.36 * (double)T * k / N;
  if (m < 0.6) return false;
  // Round 6 (tools/fm_midsize.py, profiles/r06_fm_midsize.txt): below ~2 tokens per feature the batch is small enough for the
  // token-major kernel to be latency-bound -- it reaches ~4.2 TB/s, not the 6.2 of a full batch -- and the feature-major route
  // already wins from ~1 token per feature: T = 2880 (one anyres image, m = 0.96) 2.11 -> 2.05 ms, 4096 (m = 1.36) 2.67 -> 2.56;
  // T = 2048 (m = 0.68) is a tie, 1024 stays token-major.
  const double tm_rate = m >= 2.0 ? 6.2 : 4.2;
  const double gain_ps = d * (4.0 / tm_rate - (esize + 4.0 / m) / 8.0);   // per pair
  return gain_ps >= 500.0;
}
// lanes per feature group of fm_dot_kernel: 16 when a feature has >= ~12 pairs (k = 256 at 8192 tokens: 22), else 4
inline int fm_group_lanes(int T, int k, int N) { return 1.36 * (double)T * k >= 12.0 * N ? 16 : 4; }
// the plan's side (no activation type there: the workspace is sized for the 16-bit case, run_fast asks fm_pays() again)
inline bool fm_shape_ok(int T, int k, int N, int d, int r_max) {
  int nw, lpr;
  rescore_shape(T, k, nw, lpr);
  if (lpr != 1 || r_max > 0xFFF || (long)T * r_max + (long)N * 16 >= (1L << 31)) return false;   // (fm_target's 12 bits)
  return fm_pays(T, k, N, d, 2);
}

}  // namespace
