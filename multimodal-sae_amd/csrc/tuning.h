// tuning.h -- every build-time tuning knob, ablation switch and measurement probe of libmsae_hip.so, in ONE place.
//
// The product build (csrc/build.sh) defines none of these macros: the knobs take the defaults below (each the winner of
// an A/B run recorded in NOTEBOOK.md), the ablation switches are `false`, the probes expand to nothing.  Instrumented
// builds (tools/build_dbg.sh, tools/gpu_ab*.sh) pass -D flags; the kernels test `msae_tuning::...` constants with
// `if constexpr` / plain `if`, so product sources carry no conditional compilation of their own.  Experiments that were
// measured and not kept live under tools/tuning/ (e.g. the 64-byte / 4-slot ring GEMM, gemm_mfma64.h), not here.
#pragma once

// ---- knobs (value macros, default = product) ------------------------------------------------------------------------
#ifndef MSAE_GEMM_STAGGER      // candidate GEMM: 1 = waves 4-7 issue their LDS-DMA pieces behind k-step MSAE_GEMM_STAGGER_AT
#define MSAE_GEMM_STAGGER 1    // (-4 % on the main pass with tile-major operands: profiles/r03_ab_stagger_tile_major.txt); 0 = off, 2 = odd waves
#endif
#ifndef MSAE_GEMM_STAGGER_AT   // 0 since the wave-block staging of round 5 (issue is ~100 cycles instead of ~700: the earlier the pieces leave,
#define MSAE_GEMM_STAGGER_AT 0 // the better -- 3.93-3.97 ms against 4.04-4.07 behind k-step 1, two boxes: profiles/r05_ab_stagger_at.txt)
#endif
#ifndef MSAE_GEMM_GM           // candidate GEMM: super-tile of output tiles an XCD's 32 workgroups work on together (gemm_map_tile): GM row tiles
#define MSAE_GEMM_GM 8         // x GN column tiles, GM * GN = 32.  8 x 4 fetches 12 operand blocks per k-step for 32 tiles; 4 x 8 the same with the
#define MSAE_GEMM_GN 4         // roles swapped, 16 x 2 / 2 x 16 fetch 18 (profiles/r05_ab_supertile.txt)
#endif
#ifndef MSAE_SK_UN             // weight-stream kernel: 64-B k-steps per B batch at 64 tokens (halved per doubling of the tile)
#define MSAE_SK_UN 4
#endif
#ifndef MSAE_SK_ABL            // weight-stream kernel ablations: 1 no A chunk traffic, 2 no B loads, 4 no MFMAs (results invalid)
#define MSAE_SK_ABL 0
#endif
#ifndef MSAE_RESCORE_U         // re-score: 16-B loads per lane and batch
#define MSAE_RESCORE_U 16
#endif
#ifndef MSAE_RESCORE_LPR       // lanes that share a row of W_enc in the FIRST round's re-scoring stream: 1, or 4 (64-B pieces per
#define MSAE_RESCORE_LPR 1     // row and instruction, 16 rows per pass: measured 1.61 ms against 1.17 -- not the default)
#endif
#ifndef MSAE_GUARD_ZETA        // first re-score round reaches zeta sigma below the k-th coarse value
#define MSAE_GUARD_ZETA 1.f
#endif

// ---- load flavours of the weight streams (each row piece is read exactly once per call: non-temporal by default) -----------
#ifdef MSAE_SK_PLAIN
#define MSAE_SK_LOAD(p) (*(p))
#else
#define MSAE_SK_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifdef MSAE_SK_NOFENCE
#define MSAE_SK_FENCE() do { } while (0)
#else
#define MSAE_SK_FENCE() __builtin_amdgcn_sched_barrier(0)   // a batch of loads is issued as a batch, where it is written
#endif
// The S = 1 weight stream reads every 1-KiB row piece exactly once per call: non-temporal loads (0.141 -> 0.131 ms at T = 1).
// NOT for the re-scoring rows: a lane fetches a 128-B line in eight 16-B loads and lives on the cache holding it in between
// (non-temporal there: 1.18 -> 3.34 ms, profiles/r02_ab_nontemporal.txt).
#ifdef MSAE_GEMV_PLAIN_LOADS
#define MSAE_STREAM_LOAD(p) (*(p))
#else
#define MSAE_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifdef MSAE_MF_PLAIN_LOADS
#define MSAE_MF_LOAD(p) (*(p))
#else
#define MSAE_MF_LOAD(p) __builtin_nontemporal_load(p)
#endif

#ifdef MSAE_ADAM_PLAIN_LOADS
#define MSAE_ADAM_LOAD(p) (*(p))
#define MSAE_ADAM_STORE(v, p) (*(p) = (v))
#else
#define MSAE_ADAM_LOAD(p) __builtin_nontemporal_load(p)
#define MSAE_ADAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

// ---- ablation switches (results are INVALID when one is set; stage clocks of the others stay comparable) ---------------------
namespace msae_tuning {
#ifdef MSAE_ABL_NOEPI
constexpr bool ABL_NOEPI = true;        // candidate GEMM: skip the threshold epilogue's element loop
#else
constexpr bool ABL_NOEPI = false;
#endif
#ifdef MSAE_ABL_NOFLUSH
constexpr bool ABL_NOFLUSH = true;      // ... drop the LDS queue instead of flushing it
#else
constexpr bool ABL_NOFLUSH = false;
#endif
#ifdef MSAE_ABL_NOLEAD
constexpr bool ABL_NOLEAD = true;       // ... no outlier k-tile and no acc * m - E pass in the MAIN pass (the round-5 verdict's A/B: what the tile costs)
#else
constexpr bool ABL_NOLEAD = false;
#endif
#ifdef MSAE_ABL_NOFALLBACK
constexpr bool ABL_NOFALLBACK = true;   // fused encode: no exact fallback launches
#else
constexpr bool ABL_NOFALLBACK = false;
#endif
#ifdef MSAE_RESCORE_NO_PRESELECT
constexpr bool RESCORE_PRESELECT = false;   // re-score: sort the whole candidate list instead of pre-selecting ~128 keys
#else
constexpr bool RESCORE_PRESELECT = true;
#endif
#ifdef MSAE_FULL_MAIN_PASS
constexpr bool MAIN_SKIPS_SAMPLE = false;   // main candidate pass over ALL rows (the sample rows twice): the round-2 layout
#else
constexpr bool MAIN_SKIPS_SAMPLE = true;
#endif
#ifdef MSAE_GEMM_TIMELINE
constexpr int GEMM_TIMELINE = MSAE_GEMM_TIMELINE + 0 == 2 ? 2 : 1;
#else
constexpr int GEMM_TIMELINE = 0;
#endif
}  // namespace msae_tuning

// ---- probes: s_memtime stamps (tools/gemm_timeline.py, tools/rescore_timeline.py) ------------------------------------------
// candidate GEMM: workgroup 0 / wave 0, 8 stamps per output tile into GemmEpilogue::timeline (null in the product)
#if defined(MSAE_GEMM_TIMELINE) && MSAE_GEMM_TIMELINE != 2
#define MSAE_TL(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && ep.timeline && tl_tile < 64) \
    ep.timeline[tl_tile * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSAE_TL(slot) do { } while (0)
#endif
#if defined(MSAE_GEMM_TIMELINE) && MSAE_GEMM_TIMELINE == 2   // inside k-tile 8 of every output tile (wave MSAE_TLK_WAVE)
#ifndef MSAE_TLK_WAVE
#define MSAE_TLK_WAVE 0
#endif
#define MSAE_TLK(cond, slot) do { if ((cond) && blockIdx.x == 0 && threadIdx.x == 64 * MSAE_TLK_WAVE && ep.timeline && tl_tile < 64) \
    ep.timeline[tl_tile * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSAE_TLK(cond, slot) do { } while (0)
#endif
// re-score: thread 0 of the first 64 tokens, 16 stamps each
#ifdef MSAE_RESCORE_TL
__device__ unsigned long long g_rs_tl[64 * 16];
#define MSAE_RTL(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (slot) < 16) g_rs_tl[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MSAE_RTL_VALUE(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_rs_tl[blockIdx.x * 16 + (slot)] = (v); } while (0)
#else
#define MSAE_RTL(slot) do { } while (0)
#define MSAE_RTL_VALUE(slot, v) do { } while (0)
#endif
