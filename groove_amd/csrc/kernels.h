// kernels.h — HIP kernels of the render path (gfx950 / CDNA4, wave64).
//
// Mapping: one voice (instrument kernels) or one effect channel (IIR /
// delay-line kernels) per lane; the frame axis of a block is walked sequentially inside
// the lane with all state in registers, because the oscillator phase, envelope and IIR
// recurrences serialise time.  Every per-frame global access is `base + f*n + lane`, so a
// wavefront touches 64 consecutive floats (256 B) per instruction.  Per-lane parameters and
// state live in "word-major SoA" buffers: word w of lane v at buf[w*n + v], loaded once
// per block into registers and (state only) written back once.
//
// No MFMA: nothing here is a contraction.  The instrument kernels are VALU-bound (see
// DESIGN.md §5 for the op budget), the effect and mix kernels HBM-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "dsp_core.h"
#include "derive.h"
#include "diag.h"

namespace groove {

template <class T> struct WordsOf { uint32_t w[sizeof(T) / 4]; };

// Word-major SoA access through a buffer resource: the descriptor sits in 4 SGPRs, the row
// offset (w * n * 4) is the scalar soffset and the lane offset (v * 4) one shared 32-bit VGPR,
// so loading or storing a 40-word record costs no address registers at all (with flat
// `global_load` the compiler keeps one 64-bit VGPR address per word alive from the load to
// the matching store at the end of the kernel: +80 VGPRs, one occupancy step).
// Limit: words * n * 4 bytes < 4 GiB (checked on the host when a bank is created).
template <class T>
__device__ __forceinline__ T soa_load(const uint32_t* __restrict__ buf, uint32_t n, uint32_t v) {
  static_assert(sizeof(T) % 4 == 0, "word-major SoA needs 4-byte multiples");
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(buf), 0, (int)(sizeof(T) / 4 * n * 4u), 0x00020000);
  WordsOf<T> t;
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4; ++i)
    t.w[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(v * 4u), (int)(i * n * 4u), 0);
  return __builtin_bit_cast(T, t);
}
template <class T>
__device__ __forceinline__ void soa_store(uint32_t* __restrict__ buf, uint32_t n, uint32_t v, const T& x) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)(sizeof(T) / 4 * n * 4u), 0x00020000);
  const WordsOf<T> t = __builtin_bit_cast(WordsOf<T>, x);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4; ++i)
    __builtin_amdgcn_raw_buffer_store_b32((int)t.w[i], rsrc, (int)(v * 4u), (int)(i * n * 4u), 0);
}

// Waves per SIMD each uniform-kernel kind is register-budgeted for (512 VGPRs / waves, in steps of 8).  Re-measured on
// the un-packed code (-fno-slp-vectorize), 1,000,000 voices, in-job: F32 kinds 5 -> +1.8 % in the all-voices window
// (+-0 over the project), 4 the same, 8 -> -12 %; smooth kinds 6 -> -3 %, 3 -> -10 %.
#ifndef GROOVE_WAVES_F32_STATIC
#define GROOVE_WAVES_F32_STATIC 5
#endif
#ifndef GROOVE_WAVES_F32_RETUNE
#define GROOVE_WAVES_F32_RETUNE 5
#endif
#ifndef GROOVE_WAVES_SMOOTH_STATIC
#define GROOVE_WAVES_SMOOTH_STATIC 5 /* round 5: 5 (102 VGPRs) instead of 4 for both smooth-f64 kinds — after the polynomial envelopes and the in-place phase
                                        add their bodies fit: the driver's window 0.4395 -> 0.4329 ms per block (median of five, in-job, profiles/r05_budgets_ab.log);
                                        the fp32-LFO kinds at 6 lose (0.4555 against 0.4439) */
#endif
#ifndef GROOVE_WAVES_SMOOTH_RETUNE
#define GROOVE_WAVES_SMOOTH_RETUNE 5
#endif
#ifndef GROOVE_WAVES_F64
#define GROOVE_WAVES_F64 4 /* round 6: 4 (128 VGPRs) instead of 2, with class-specialised bodies for these kinds too.  Not for their own occupancy — they
                              are a few workgroups — but so that they can be PLACED: beside the mix kernel's five waves per SIMD (5 x 96 of 512 registers) a wave
                              of 133+ registers only found room where a CU was draining at the end of a mix launch, and the kind's kernel took ~405 us per
                              block beside launches of ~345 (227 us alone): the library-proportioned bank's long pole with 1.9 % of its voices.  At 128 a
                              workgroup fits as soon as ONE mix workgroup has left its CU. */
#endif
#ifndef GROOVE_WAVES_ANY
#define GROOVE_WAVES_ANY 4 /* the all-kinds kernel of banks that do not fill the chip.  Round 3: 4 (128 VGPRs) instead of 3 — in-job, blocks 5-24:
                              250,000 voices 0.212 -> 0.197 ms per block, 500,000 0.350 -> 0.331, 125,000 unchanged (5 is worse below 500,000:
                              0.172 at 125,000); the two exact-f64 base kinds, whose bodies need 133 VGPRs, no longer run in this kernel (the
                              host gives their workgroups to the per-kind kernels) */
#endif
/* (A second copy of the frame loop for waves whose 64 lanes all sound — no exec-mask region round the frame, no two moves of zero in front of
   it — was measured twice in round 6 and lost twice: in the shared bodies 0.523 against 0.350 ms per block (the doubled loops spilled), inside
   the FAST copies 0.3240 - 0.3258 against 0.3138 - 0.3174.  Not in the source any more.) */
#ifndef GROOVE_FAST_TABLE_LOOP
#define GROOVE_FAST_TABLE_LOOP 1 /* segments whose look-aheads are all up run a frame loop of their own, compiled without the flag tests (run_frames_segmented `fast`):
                                    1 in the F32 kinds' fp32-filter bodies; 2 also in the smooth-f64 kinds' fp32-filter bodies with a sine / triangle LFO (no scratch access in
                                    either loop, and nothing on the clock: 0.3367 - 0.3419 against 0.3344 - 0.3383 ms per block); 3 also in the F32 kinds' f64-filter bodies (the
                                    library-proportioned bank 0.348 -> 0.387: their other loop spills); 0: one loop everywhere.  One job each, tools/ab_bench.sh. */
#endif
#ifndef GROOVE_AMP_IN_TABLE
#define GROOVE_AMP_IN_TABLE 0 /* 1: the retuned kinds' table entries carry the amplitude envelope's value of the frame too (TabLayout::kAmp, welsh_frame's AMPTAB: the
                                 table flag then also asks for a shared amplitude stage).  Three vector instructions a table frame less — and nothing on the clock, three times:
                                 with a flag and a branch of its own before the frame loop ran in chunks (0.362 - 0.373 against 0.354 - 0.357 ms per block), and riding
                                 on the coefficient flag after (0.3464 - 0.3512 against 0.3483 - 0.3502; library 0.3469 - 0.3565 against 0.3494 - 0.3533), and with the FAST copies, whose promise then
                                 covers the amplitude envelope's record too (0.3175 - 0.3339 against 0.3098 - 0.3138); one job each, tools/ab_bench.sh.  The frame is not short of issue slots at that point; left in the source, off. */
#endif
#ifndef GROOVE_LFO_LOOKAHEAD
#define GROOVE_LFO_LOOKAHEAD 1 /* the smooth-f64 kinds' LFO look-ahead (below, "LFO look-ahead"); 0: every lane advances its LFO's recurrences (A/B builds) */
#endif
#ifndef GROOVE_COEF_LOOKAHEAD
#define GROOVE_COEF_LOOKAHEAD 1 /* the retuned kinds' coefficient look-ahead (below, "coefficient look-ahead"); 0: every lane retunes for itself (round 5's code, for A/B builds) */
#endif
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
// Which entry of the kind-sorted workgroup list (cheapest base kind first) workgroup blockIdx.x of a launch takes: from the END.
// The most expensive kinds start first (longest-processing-time-first), so that a launch of more than one round of workgroups does
// not end with its longest workgroups alone on the chip.  Same results (a workgroup's voices do not care when they run); measured
// against the forward order in one job, blocks 5-24 (profiles/r04_wg_order_ab.log): 250,000 voices 0.188 / 0.183 -> 0.177 / 0.181 ms
// per block, 350,000 0.242 / 0.246 -> 0.230 / 0.235, 500,000 0.302 / 0.319 -> 0.290 / 0.290; 125,000, 1,000,000 and config #5 unchanged.
#define GROOVE_WG_SLOT(n_wgs) ((n_wgs) - 1u - blockIdx.x)

// ------------------------------------------------------------------ wave-uniform parameters
// Voices of one synth share a patch (the reference's WelshSynth is one patch + a voice
// store), so in a well-grouped project every lane of a wavefront carries identical
// parameter words.  Then the whole parameter struct is moved to SGPRs with
// v_readfirstlane: ~35 VGPRs freed, and every `switch (waveform)` / routing test below
// becomes a scalar branch instead of an exec-mask region.
template <class T>
__device__ __forceinline__ T make_scalar(const T& x) {
  WordsOf<T> w = __builtin_bit_cast(WordsOf<T>, x);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4; ++i) w.w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.w[i]);
  return __builtin_bit_cast(T, w);
}

// groove_bank_render_mix_deferred: the PREVIOUS block's bus reduction rides in this launch.  A small bank's fused step is two
// launches — the render, and a reduction of its <= 64 partial rows that is all launch and latency (4 us of config #2's 16) — and
// the next block's render is on the same stream right behind: its workgroups each add up a slice of the previous block's rows
// (wavefront 0, a few lanes: one column each, all rows in flight, fixed order) while their own parameter loads are under way.
struct TpPrev { const float* rows = nullptr; float* bus = nullptr; uint32_t n_rows = 0, frames = 0; int accumulate = 0; };
__device__ __forceinline__ void tp_reduce_prev(const TpPrev& pv, uint32_t tid, uint32_t wg, uint32_t n_wg) {
  if (!pv.rows || tid >= 64u) return; // wavefront 0
  const uint32_t cols = 2 * pv.frames, per = (cols + n_wg - 1) / n_wg; // columns of this workgroup
  // CP columns per pass, 64 / CP lanes per column, eight rows per lane and batch: a bank of R <= 512 workgroups left R rows and
  // gives every workgroup ceil(512 / R) columns — all of a column's rows are in flight at once, one round trip per pass
  const uint32_t cp = per <= 1 ? 1u : (per <= 2 ? 2u : (per <= 4 ? 4u : 8u)), lpc = 64u / cp;
  const uint32_t g = tid % lpc, cl = tid / lpc; // lane = (column of the pass, group of rows)
  for (uint32_t c0 = 0; c0 < per; c0 += cp) {
    const uint32_t c = wg * per + c0 + cl;
    const bool ok = c0 + cl < per && c < cols;
    float acc = 0.0f;
    for (uint32_t rb = 0; rb < pv.n_rows; rb += lpc * 8) {
      float t[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t r = rb + g * 8 + k;
        t[k] = (ok && r < pv.n_rows) ? pv.rows[(size_t)r * cols + c] : 0.0f; // rows[workgroup][ch][frame]: column c = ch * frames + f
      }
      acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (uint32_t m = 1; m < lpc; m <<= 1) acc += __shfl_xor(acc, (int)m, 64); // the column's lanes (a fixed order)
    if (ok && g == 0) {
      const uint32_t ch = c / pv.frames, f = c % pv.frames;
      float* o = pv.bus + (size_t)f * 2 + ch;
      *o = pv.accumulate ? *o + acc : acc;
    }
  }
}

// ------------------------------------------------------------------ fused mix-bus epilogue
// Orchestrator::gather_audio's "sum += entity.value()" (orchestrator.rs:397-410) without
// materialising the voice block: every frame a lane parks its (L, R) in an LDS tile, every eight frames the
// workgroup turns the tile and writes one row segment per channel (FusedAcc below):
//     partial[workgroup][ch][frame]
// A second, tiny kernel pair sums the partial rows over workgroups (deterministic order).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
  // x + (x moved by the DPP pattern); lanes of rows outside ROW_MASK receive 0
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
// Sum over the 64 lanes; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_lane63(float x) {
  x = dpp_add<0xb1, 0xf>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4e, 0xf>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x124, 0xf>(x);  // row_ror:4
  x = dpp_add<0x128, 0xf>(x);  // row_ror:8   → every lane of a 16-lane row holds the row sum
  x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}
// LDS-transposed form.  Every frame each lane drops its (L, R) pair into a
// [8 frames][256 lanes] float2 tile with one ds_write_b64; every 8 frames the workgroup turns
// the tile: 32 lanes per frame row, each sums 8 pairs (conflict-free 256-byte row segments),
// then 5 DPP steps finish the 32-lane sum and two lanes per wave store the row totals.
// ≈ 5 VALU + 1 LDS write per frame (an all-DPP form without LDS was 17 VALU per frame and 1.5 % slower: docs/HISTORY.md);
// 16 KiB of LDS per workgroup.
struct FusedAcc {
  static constexpr uint32_t kChunk = 8;
  uint32_t prow;   // this workgroup's row pair in partial[]
  __device__ __forceinline__ explicit FusedAcc(uint32_t row) : prow(row) {}
  __device__ __forceinline__ float2* tile() {
    __shared__ float2 t[kChunk][kThreads];
    return &t[0][0];
  }
  __device__ __forceinline__ void add(float L, float R, uint32_t f) {
    tile()[(f & (kChunk - 1)) * kThreads + threadIdx.x] = make_float2(L, R);
  }
  __device__ __forceinline__ void flush(float* __restrict__ partial, uint32_t frames, uint32_t f0, uint32_t count) {
    __syncthreads();
    const uint32_t row = threadIdx.x >> 5, col = threadIdx.x & 31u; // 32 lanes per frame row
    const float2* __restrict__ src = tile() + row * kThreads + col;
    float l = 0.0f, r = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < kThreads / 32; ++j) { const float2 v = src[j * 32]; l += v.x; r += v.y; }
    // 32-lane sums: quad, quad, ror4, ror8 (16-lane row sums), then row_bcast:15 into rows 1 and 3
    l = dpp_add<0xb1, 0xf>(l); r = dpp_add<0xb1, 0xf>(r);
    l = dpp_add<0x4e, 0xf>(l); r = dpp_add<0x4e, 0xf>(r);
    l = dpp_add<0x124, 0xf>(l); r = dpp_add<0x124, 0xf>(r);
    l = dpp_add<0x128, 0xf>(l); r = dpp_add<0x128, 0xf>(r);
    l = dpp_add<0x142, 0xa>(l); r = dpp_add<0x142, 0xa>(r);
    if (col == 31 && row < count) { // lanes 31 and 63 of each wave hold the totals of their rows
      partial[((size_t)prow * 2 + 0) * frames + f0 + row] = l;
      partial[((size_t)prow * 2 + 1) * frames + f0 + row] = r;
    }
    __syncthreads();
  }
};
// The planar block's stores.  (Non-temporal stores and whole 1 KB rows out of the bus tile were both measured in round 3 and
// lost to this plain form: docs/HISTORY.md.  So did, in round 4, a STORE WAVE — a fifth wavefront per workgroup that drains a
// double-buffered tile, so that the four voice wavefronts issue no store, no address arithmetic and no tile turn: the
// materialised million-voice window 0.885 - 0.891 ms per block against 0.790 - 0.796 in one job, profiles/r04_store_wave_ab.log.
// The stores are not what the voice waves wait for; a fifth wave per workgroup is a fifth of the wave slots.)
__device__ __forceinline__ void block_store(float* __restrict__ p, float x) { *p = x; }
// (Round 6 tried the two stores through a BUFFER resource — base in SGPRs, scalar row offset, one 32-bit lane offset, no 64-bit address
// arithmetic per frame: the materialised million-voice window 0.899 ms per block against 0.736, profiles/r06_materialised_stores_ab.log.
// The same log has the counters: the waves of this form wait three times as long as the fused form's, on the memory pipeline, not on issue.)
// Shared frame loop of the instrument kernels: `frame(f, L, R)` computes one frame of this
// lane's voice; the epilogue either stores the planar block or feeds the fused bus sum.
// FUSED: only the rows of the fused bus sum (partial[workgroup][ch][frame]) are produced.  Otherwise the planar block is
// stored AND the same rows are written to `rows`: the block's own lane sums, which groove_mix then reduces instead of
// reading the 8 bytes per voice-frame back (a block keeps them valid until something else writes it).
template <bool FUSED, class FrameFn>
__device__ __forceinline__ void run_frames(uint32_t frames, uint32_t n, uint32_t v, bool active, size_t ch_stride,
                                           float* __restrict__ out, float* __restrict__ rows, uint32_t prow, FrameFn&& frame) {
  FusedAcc acc(prow);
  constexpr uint32_t C = FusedAcc::kChunk;
  for (uint32_t f = 0; f < frames; ++f) {
    float L, R;
    frame(f, L, R);
    acc.add(active ? L : 0.0f, active ? R : 0.0f, f);
    if ((f & (C - 1)) == C - 1) acc.flush(rows, frames, f - (C - 1), C);
    if (!FUSED && active) { // uniform row base + 32-bit lane offset
      float* __restrict__ rowL = out + (size_t)f * n;
      float* __restrict__ rowR = out + ch_stride + (size_t)f * n;
      block_store(rowL + v, L);
      block_store(rowR + v, R);
    }
  }
  if (frames & (C - 1)) acc.flush(rows, frames, frames & ~(C - 1), frames & (C - 1));
}
template <int V> struct IntTag { static constexpr int value = V; };
// Minimum of x over the 64 lanes (wave-uniform result).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
  auto step = [](uint32_t v, auto ctrl) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, decltype(ctrl)::value, 0xf, 0xf, false);
    return o < v ? o : v;
  };
  x = step(x, IntTag<0xb1>{});   // quad_perm [1,0,3,2]
  x = step(x, IntTag<0x4e>{});   // quad_perm [2,3,0,1]
  x = step(x, IntTag<0x124>{});  // row_ror:4
  x = step(x, IntTag<0x128>{});  // row_ror:8   -> every lane of a 16-lane row holds the row minimum
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
  return min(min(r0, r1), min(r2, r3));
}
// Frame 0 by `first` (checked form); then boundary-free SEGMENTS (dsp_core.h): `begin(live)` handles the
// envelope boundaries due now and returns the frames this lane can run unchecked, the wave takes the
// minimum over its ACTIVE lanes, and `live_frame` runs that many frames without boundary or idle tests.
// `begin` also prepares the segment (float stage counters), live frames leave the envelope counters
// alone, and `end(seg, live)` moves them once per segment (welsh_segment_end_hoisted); lanes that are not `active`
// are never live, so their L and R stay zero without a select per frame.
//
// Why ACTIVE lanes only (the stall of rounds 2-3).  A lane that is not active carries a SHADOW of some
// voice's state record so that its loads have an address: in a partly filled wave its own wave's first voice, in a PADDING
// wave (count 0: welsh_upload_params fills every kind's last workgroup up to four waves) the first voice of the kind's first
// wave — a voice that another workgroup of the SAME launch owns, and stores at the end of its block while this workgroup, in a
// later round of the grid, may be loading it.  The shadow can therefore be a torn record: after a note-on an instant-attack
// voice holds (ATTACK, n 0, N 0), one block later (SUSTAIN, n, N 2^32-1), and the mixture (SUSTAIN, N 0) is a plateau that is
// "at its boundary" on every frame: frames-to-boundary 0, for ever.  With the shadow lanes in the minimum that was a segment
// of zero frames — the endless loop of rounds 2 and 3 (one workgroup per ~10^5, never a wrong sample: shadows are never
// stored or summed).  Shadow lanes now contribute 2^32-1.  `on_zero(f, mine, wmin)` is called (wave-uniformly) if the minimum
// is 0 all the same — it counts (diag.h) — and the segment is then one frame, which is what the checked form would do.
// `setup(live, seg)` runs once per segment (wave-uniformly: every lane of the wave is in it) once the segment's length is known, and
// `pre(k)` before frames k, k + 64, ... of the segment, outside the `live` test: the look-ahead tables' fill (welsh_block).
// `fast()` (wave-uniform, asked once per segment after `setup`): every look-ahead this kind has is up for the segment — its frames then
// run `fast_frame`, the frame compiled with the table flags as constants (dsp_core.h welsh_frame's TABS), in a loop of their own: no
// flag tests, no taken branches round the code they guard.  (Why it matters: the instruction cache.  SQC_ICACHE_BUSY_CYCLES / SQ_CYCLES
// of the million-voice window was 0.85 with the flags tested in the frame — tools/diag_pmc.sh —: forty wavefronts per cache fetch a
// stream in which every tenth instruction is a branch, half of them taken.)
template <bool FUSED, bool FASTONLY = false, class FirstFn, class BeginFn, class SetupFn, class PreFn, class LiveFn, class FastFn, class FastFrameFn, class EndFn, class ZeroFn>
__device__ __forceinline__ void run_frames_segmented(uint32_t frames, uint32_t n, uint32_t v, bool active, size_t ch_stride,
                                                     float* __restrict__ out, float* __restrict__ rows, uint32_t prow, FirstFn&& first, BeginFn&& begin,
                                                     SetupFn&& setup, PreFn&& pre, LiveFn&& live_frame, FastFn&& fast, FastFrameFn&& fast_frame, EndFn&& end, ZeroFn&& on_zero) {
  if (frames == 0) return;
  constexpr uint32_t C = FusedAcc::kChunk;
  static_assert(C > 1, "frame 0 never completes a chunk");
  FusedAcc acc(prow);
  auto put = [&](uint32_t f, float L, float R) {
    acc.add(L, R, f);
    if ((f & (C - 1)) == C - 1) acc.flush(rows, frames, f - (C - 1), C);
    if (!FUSED && active) {
      block_store(out + (size_t)f * n + v, L);
      block_store(out + ch_stride + (size_t)f * n + v, R);
    }
  };
  {
    float L, R;
    first(L, R);
    put(0, active ? L : 0.0f, active ? R : 0.0f);
  }
  uint32_t f = 1;
  while (f < frames) {
    bool live;
    const uint32_t mine = begin(live);
    live = live && active;
#ifdef GROOVE_DIAG_SHADOW_IN_MIN /* diag.h: round 3's minimum, shadows included */
    const uint32_t wmin = wave_min_u32(mine);
#else
    const uint32_t wmin = wave_min_u32(active ? mine : 0xFFFFFFFFu);
#endif
    if (wmin == 0) on_zero(f, mine);
    const uint32_t seg = max(1u, min(wmin, frames - f));
    setup(live, seg);
    // in chunks of CoefTab::kFrames frames: `pre` (the look-ahead tables' fill: long, cold code) stays out of the frame loop proper
    const bool all_tables = FASTONLY ? true : fast();
    for (uint32_t k0 = 0; k0 < seg; k0 += 64u) {
      pre(k0);
      const uint32_t k1 = min(seg, k0 + 64u);
      if (FASTONLY || all_tables) {
        for (uint32_t k = k0; k < k1; ++k, ++f) {
          float L = 0.0f, R = 0.0f;
          if (live) fast_frame(k, L, R);
          put(f, L, R);
        }
      } else if constexpr (!FASTONLY) {
        for (uint32_t k = k0; k < k1; ++k, ++f) {
          float L = 0.0f, R = 0.0f;
          if (live) live_frame(k, L, R);
          put(f, L, R);
        }
      }
    }
    end(seg, live);
  }
  if (frames & (C - 1)) acc.flush(rows, frames, frames & ~(C - 1), frames & (C - 1));
}
// Where a wave is, for the diagnostics of diag.h.
struct DiagWhere { uint32_t* diag; uint32_t wg, wave, count; };
// run_frames_segmented's `on_zero` for a Welsh wave: the guard's counter; in the GROOVE_DIAG_SHADOW_IN_MIN build also who it was.
__device__ __forceinline__ void welsh_diag_zero(const DiagWhere& dw, const WelshState& s, bool active, uint32_t f, uint32_t mine) {
  if (!dw.diag) return;
#ifdef GROOVE_DIAG_SHADOW_IN_MIN
  DiagCounters* dc = reinterpret_cast<DiagCounters*>(dw.diag);
  const bool active_zero = __any(active && mine == 0);
  if ((threadIdx.x & 63u) == 0) atomicAdd(active_zero ? &dc->zero_segments : &dc->shadow_zero_waves, 1u);
  if (mine == 0) {
    atomicAdd(&dc->shadow_zero_lanes, 1u);
    const uint32_t slot = atomicAdd(&dc->records, 1u);
    if (slot < kDiagRecords)
      dc->rec[slot] = DiagRecord{dw.wg, dw.wave, threadIdx.x & 63u, active ? 1u : 0u, dw.count, f, s.amp.state, s.amp.n, s.amp.N, s.fil.state, s.fil.n, s.fil.N};
  }
#else
  (void)s; (void)active; (void)f; (void)mine;
  if ((threadIdx.x & 63u) == 0) diag_count_zero_segment(dw.diag);
#endif
}

// ------------------------------------------------------------------ instruments
// a5 WelshVoice: Ticks::tick(frames) + Generates::generate_batch_values.
// Frame 0 is peeled (first-tick flag); RETUNE=false variants keep the filter coefficients
// loop-invariant so their f64 widening is hoisted out of the frame loop.
// ------------------------------------------------------------------ coefficient look-ahead (round 6)
// The retune of an envelope-swept filter — cutoff percent from the filter envelope's value, one exp2, the tangent polynomial, two
// reciprocals, the quotients and their widening — is ~45 of a retuning frame's ~75 vector instructions, and every lane of a wave
// whose voices were struck together computes THE SAME numbers: the cutoff depends on the patch (wave-uniform) and on the filter
// envelope, which knows nothing of the key.  There is no scalar float unit to do it once per wave; there are 64 lanes, and the
// envelope is a closed form in the stage's frame counter: when a segment starts and all live lanes of the wave agree on the filter
// envelope's stage (counter, start level, shape constants), lane j computes the coefficients of frame k + j of the segment, 64 frames
// in one pass, into a table of the wave's own in LDS, and every frame then takes its six coefficients with broadcast reads.  Same
// expressions on the same values as the per-lane path (welsh_frame_front / lp24_t_from_pct / lp24_coef*_from_t): the same bits.
// Waves whose voices started apart and segments shorter than eight frames keep the per-lane path.  (LFO-swept cutoffs did too until the
// LFO look-ahead below.)
//
// LFO look-ahead (round 6, the smooth-f64 kinds: LFO on the pitch or the pulse width).  A voice's LFO runs while the voice sounds, so
// voices struck together share its phase as well, and what it does to the oscillators' edges — 2^(l depth) or l depth — is one number
// per wave and frame that every lane was advancing with ~13 f64 operations a frame (a rotation, a 4-term exponential series).  Same
// table, one more double per entry: lane j evaluates it EXACTLY from the phase of frame k + j (the exact-f64 kind's expressions:
// welsh_lfo_mod_exact), the frames read it, the phase moves once per segment and the recurrences are re-seeded there for whatever
// segment has to run lane by lane later.  Not the recurrences' bits (they differ from the exact evaluation by their random walk of
// ~1e-16 a frame, re-seeded every block): tests/test_gpu_library.py compares the two paths at 2e-6 and both with the oracle.
struct CoefTab {
  static constexpr uint32_t kFrames = 64, kMinSegment = 8;
  // One entry per frame: the six coefficients (f64: 48 bytes; f32 in the fp32-filter bodies: 24) where the kind retunes, then `mod`
  // (8 bytes) in the smooth-f64 kinds; an fp32 entry without `mod` is padded to 32 bytes (a power of two: its address is then a scalar
  // AND and one vector add instead of a 64-bit multiply-add per frame).  3.5 KiB per wave at most.
  static constexpr uint32_t kMaxEntry = 56; // (64 would take the per-kind kernels' LDS over a fifth of the CU's: four workgroups per CU instead of five)
  static __device__ __forceinline__ double* wave_base() {
    __shared__ double t[kWaves][kFrames][kMaxEntry / 8];
    return &t[threadIdx.x >> 6][0][0];
  }
  // T at byte `off` of frame i's entry (entries `stride` bytes apart), read through an LDS (address space 3) pointer built from a 32-bit
  // byte address.  Word by word: a struct cannot be assigned across address spaces; the compiler merges the words into wide ds_read /
  // ds_write.
  typedef __attribute__((address_space(3))) uint32_t* LdsWords;
  template <class T> static __device__ __forceinline__ T load(uint32_t i, uint32_t stride, uint32_t off = 0) {
    typedef __attribute__((address_space(3))) char* LdsBytes;
    // (The address is wave-uniform; forcing it through SGPRs — two v_readfirstlane, scalar multiply and add, a move — was measured and
    // lost to this per-lane form, whose index arithmetic the compiler does with one v_mad_u64_u32: 0.3594 - 0.3676 against 0.3516 - 0.3538
    // ms per block in one job, tools/ab_bench.sh, round 6.)
    const LdsWords w = (LdsWords)(uintptr_t)((uint32_t)(uintptr_t)(LdsBytes)wave_base() + i * stride + off);
    WordsOf<T> t;
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 4; ++k) t.w[k] = w[k];
    return __builtin_bit_cast(T, t);
  }
  // ... and into lane j's own entry (the fill: a per-lane address)
  template <class T> static __device__ __forceinline__ void store_of_lane(uint32_t j, uint32_t stride, uint32_t off, const T& x) {
    typedef __attribute__((address_space(3))) char* LdsBytes;
    const LdsWords w = (LdsWords)(uintptr_t)((uint32_t)(uintptr_t)(LdsBytes)wave_base() + j * stride + off);
    const WordsOf<T> t = __builtin_bit_cast(WordsOf<T>, x);
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 4; ++k) w[k] = t.w[k];
  }
};
// The layout of a kind's entries: [coefficients][LFO: `mod` (f64, smooth kinds) or the fp32 value][amplitude envelope's value (fp32)].
// The amplitude value rides along where the entry stays within kMaxEntry (not the smooth-f64 kinds' f64-filter bodies: five workgroups'
// tables and bus tiles fill a CU's 160 KiB of LDS to within 10 KiB).
template <bool F32, bool COEF, bool LFO, bool SMOOTH> struct TabLayout {
  static constexpr uint32_t kCoef = COEF ? (F32 ? (uint32_t)sizeof(Lp24CoefF) : (uint32_t)sizeof(Lp24CoefD)) : 0u;
  static constexpr uint32_t kModOff = kCoef;
  static constexpr uint32_t kLfo = LFO ? (SMOOTH ? 8u : 4u) : 0u;
  static constexpr uint32_t kAmpOff = kCoef + kLfo;
  static constexpr bool kAmp = COEF && GROOVE_AMP_IN_TABLE && kAmpOff + 4u <= CoefTab::kMaxEntry;
  static constexpr uint32_t kRaw = (kAmpOff + (kAmp ? 4u : 0u) + 7u) & ~7u;
  static constexpr uint32_t kStride = F32 && COEF && kRaw <= 32u ? 32u : kRaw; // (a power of two where it costs nothing)
  static_assert(kStride <= CoefTab::kMaxEntry && kCoef % 8 == 0, "entry layout"); // (stride 0: a kind without a table)
};
// Do the live lanes of this wave share the filter envelope's stage (-> tab)?  The LFO's phase (-> ltab)?  Then their description,
// from the first live lane, in SGPRs.  tab / ltab: 0 / 1 the compiler KNOWS to be in an SGPR (readfirstlane).
// cut: the coefficients follow the LFO (an LFO-swept cutoff in an F32 kind), not the filter envelope.
struct WaveUniform { float A, c1, c2, tf; float aA, a1, a2, ta; uint32_t lph_lo, lph_hi; uint32_t tab, ltab, cut; };
__device__ __forceinline__ float lane_value(float x, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane)); }
__device__ __forceinline__ bool same_bits(float a, float b) { return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b); }
template <bool COEF, bool LFO, int LFO_MODE, int CL, bool AMP>
__device__ __forceinline__ WaveUniform wave_uniform(const WelshParams& p, const WelshState& s, const WelshScratch& sc, bool live, uint32_t look) {
  WaveUniform u{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u, 0u, 0u};
  const uint64_t mask = __ballot(live);
  if (mask == 0) return u;
  const int l0 = __builtin_ctzll(mask); // wave-uniform
  if constexpr (LFO) {
    // (a noise LFO has a state of its own per voice: class OSC_ANY of an F32 kind can be one)
    if ((look & 2u) && osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u) != (uint32_t)GROOVE_WAVE_NOISE) {
      const uint32_t lo = (uint32_t)s.lfo.phase, hi = (uint32_t)(s.lfo.phase >> 32);
      u.lph_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, l0); u.lph_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, l0);
      u.ltab = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(live && (lo != u.lph_lo || hi != u.lph_hi)) == 0 ? 1 : 0);
    }
  }
  if constexpr (COEF) {
    if constexpr (LFO && LFO_MODE == LFO_F32) { // an LFO-swept cutoff: the coefficients are the shared LFO's (welsh_frame_front: the envelope has precedence)
      if (!(p.flags & WF_RETUNE_ENV) && (p.flags & WF_LFO_CUTOFF) && (look & 1u) && u.ltab != 0u) { u.tab = 1u; u.cut = 1u; }
    }
    if ((p.flags & WF_RETUNE_ENV) && (look & 1u)) {
      u.A = lane_value(s.fil.A, l0); u.c1 = lane_value(sc.fc1, l0); u.c2 = lane_value(sc.fc2, l0); u.tf = lane_value(sc.tf, l0);
      // bit patterns, so that a NaN (never produced; a torn shadow record could hold anything, but shadows are not live) cannot fake agreement
      const bool same = same_bits(s.fil.A, u.A) && same_bits(sc.fc1, u.c1) && same_bits(sc.fc2, u.c2) && same_bits(sc.tf, u.tf);
      u.tab = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(live && !same) == 0 ? 1 : 0);
    }
    if constexpr (AMP) { // the entries carry the amplitude envelope's value: its stage must be shared as well
      if (u.tab != 0u) {
        u.aA = lane_value(s.amp.A, l0); u.a1 = lane_value(sc.ac1, l0); u.a2 = lane_value(sc.ac2, l0); u.ta = lane_value(sc.ta, l0);
        const bool same = same_bits(s.amp.A, u.aA) && same_bits(sc.ac1, u.a1) && same_bits(sc.ac2, u.a2) && same_bits(sc.ta, u.ta);
        u.tab = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(live && !same) == 0 ? 1 : 0);
      }
    }
  }
  return u;
}
// Lane j: the entry of frame k0 + j of the segment, into the wave's table.  (All 64 lanes take part, live or not: the inputs are
// wave-uniform.)
template <bool F32, bool COEF, bool LFO, int LFO_MODE, int CL>
__device__ __forceinline__ void wave_tab_fill(const WelshParams& p, const RenderConsts& rc, const WaveUniform& u, uint32_t k0) {
  typedef TabLayout<F32, COEF, LFO, LFO_MODE == LFO_F64_SMOOTH> Lay;
  const uint32_t j = threadIdx.x & 63u;
  float lfo = 0.0f;
  if constexpr (LFO) {
    if (u.ltab != 0u) { // frame k of the segment evaluates the LFO at the segment's start phase + (k + 1) increments
      const uint64_t ph = (((uint64_t)u.lph_hi << 32) | u.lph_lo) + (uint64_t)(k0 + j + 1u) * p.lfo_inc;
      if constexpr (LFO_MODE == LFO_F64_SMOOTH) CoefTab::store_of_lane(j, Lay::kStride, Lay::kModOff, welsh_lfo_mod_exact<CL>(p, ph));
      else { lfo = welsh_lfo_value_f32<CL>(p, ph); CoefTab::store_of_lane(j, Lay::kStride, Lay::kModOff, lfo); }
    }
  }
  if constexpr (COEF) {
    if (u.tab != 0u) {
      float pct;
      if (LFO && LFO_MODE == LFO_F32 && u.cut != 0u) pct = welsh_lfo_cutoff_pct(p, lfo);
      else pct = welsh_env_cutoff_pct(p, env_shape(u.tf + (float)(k0 + j), u.A, u.c1, u.c2)); // the hoisted frames' counter sc.tf grows by 1.0f a frame (exact below 2^24)
      bool hi;
      const float t = lp24_t_from_pct(pct, rc, hi);
      if constexpr (F32) CoefTab::store_of_lane(j, Lay::kStride, 0, lp24_coeff_from_t(p.fc, t, hi));
      else CoefTab::store_of_lane(j, Lay::kStride, 0, lp24_coefd_from_t(p.fc, t, hi, (p.flags & WF_COEF_WIDE) != 0));
      if constexpr (Lay::kAmp) CoefTab::store_of_lane(j, Lay::kStride, Lay::kAmpOff, env_shape(u.ta + (float)(k0 + j), u.aA, u.a1, u.a2));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the wave's own reads below come after these writes
  __builtin_amdgcn_wave_barrier();
}

template <bool FUSED, bool RETUNE, int LFO_MODE = LFO_F64, bool UNIFORM = false, int C1 = OSC_ANY, int C2 = OSC_ANY, int CL = OSC_ANY, bool REST = false, bool F32OK = false, bool FASTONLY = false>
__device__ __forceinline__ void welsh_block(const WelshParams& p, WelshState& s, const RenderConsts& rc,
                                            uint32_t frames, uint32_t n, uint32_t v, bool active,
                                            size_t ch_stride, float* __restrict__ out, float* __restrict__ rows, uint32_t prow, const DiagWhere& dw = DiagWhere{nullptr, 0, 0, 0}) {
  WelshScratch sc = welsh_scratch_init(p, rc);
  // Static cutoff + wave-uniform patch: the six f64 coefficients are the same in every lane and
  // never change, so they ride in SGPRs (12 VGPRs back; f64 FMAs take one scalar operand).
  if constexpr (UNIFORM && F32OK && LFO_MODE != LFO_F64) {
    // F32OK: the copy of the block for workgroups whose patches carry WF_FILTER_F32 (the host builds workgroups that are uniform in
    // it and lists them in UniformArgs::wg_f32) — the same segmented block with the filter's recurrence in fp32 (dsp_core.h "fp32
    // recurrence"); the state goes back into the f64 fields of the record.  A function of its own per class triple, like the f64
    // copies: both forms in ONE function (a scalar branch per block) spilled 100 - 150 bytes per lane in their hot loops and gave
    // back two thirds of the gain (profiles/r05_f32_filter.log).
    welsh_scratch_f32_begin(p, s, rc, sc);
    if (!RETUNE) sc.coef_f = make_scalar(sc.coef_f);
    constexpr bool COEF_LA = RETUNE && GROOVE_COEF_LOOKAHEAD;
    constexpr bool LFO_LA = LFO_MODE != LFO_F64 && CL != LFO_UNUSED && GROOVE_LFO_LOOKAHEAD;
    typedef TabLayout<true, COEF_LA, LFO_LA, LFO_MODE == LFO_F64_SMOOTH> Lay;
    WaveUniform fu{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u, 0u, 0u}; // this segment's look-aheads (fu.tab, fu.ltab are wave-uniform)
    run_frames_segmented<FUSED, FASTONLY && (COEF_LA || LFO_LA)>(
        frames, n, v, active, ch_stride, out, rows, prow,
        [&](float& L, float& R) { welsh_frame<true, RETUNE, LFO_MODE, C1, C2, CL, false, REST, false, true>(p, s, rc, sc, L, R); },
        [&](bool& live) { const uint32_t k = welsh_segment_begin(p, s, live); welsh_segment_start_hoisted(s, sc); return k; },
        [&](bool live, uint32_t seg) {
          fu.tab = 0u; fu.ltab = 0u;
          if constexpr (COEF_LA || LFO_LA) { if (FASTONLY || seg >= CoefTab::kMinSegment) fu = wave_uniform<COEF_LA, LFO_LA, LFO_MODE, CL, Lay::kAmp>(p, s, sc, live, rc.look); }
          if constexpr (FASTONLY && (COEF_LA || LFO_LA)) { // the promise this copy was chosen on (welsh_wave_tables_up): counted if it does not hold
            if (!((!COEF_LA || fu.tab != 0u) && (!LFO_LA || fu.ltab != 0u)) && __ballot(live) != 0 && (threadIdx.x & 63u) == 0) diag_count_fast_table_miss(dw.diag);
          }
        },
        [&](uint32_t k) { if constexpr (COEF_LA || LFO_LA) { if ((fu.tab | fu.ltab) != 0u && (k & (CoefTab::kFrames - 1)) == 0) wave_tab_fill<true, COEF_LA, LFO_LA, LFO_MODE, CL>(p, rc, fu, k); } },
        [&](uint32_t k, float& L, float& R) {
          uint32_t tab = 0u, ltab = 0u;
          double mod = 0.0;
          float tamp = 0.0f;
          if constexpr (COEF_LA) {
            tab = fu.tab;
            if (tab != 0u) {
              sc.coef_f = CoefTab::load<Lp24CoefF>(k & (CoefTab::kFrames - 1), Lay::kStride);
              if constexpr (Lay::kAmp) tamp = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kAmpOff);
            }
          }
          float tlfo = 0.0f;
          if constexpr (LFO_LA) {
            ltab = fu.ltab;
            if constexpr (LFO_MODE == LFO_F64_SMOOTH) { if (ltab != 0u) mod = CoefTab::load<double>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); }
            else { if (ltab != 0u && ((p.flags & WF_LFO_AMP) || tab == 0u)) tlfo = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); } // (a cutoff-only LFO is in the coefficients already)
          }
          welsh_frame<false, RETUNE, LFO_MODE, C1, C2, CL, true, REST, true, true, Lay::kAmp>(p, s, rc, sc, L, R, tab, ltab, mod, tlfo, tamp);
        },
        [&]() { // fast(): every look-ahead of this kind is up.  (The F32 kinds' fp32-filter bodies only: in the others the second loop costs
                // registers — 2 - 20 scratch accesses per frame in one loop or the other, and 0.3438 -> 0.3607 ms per block with it everywhere.)
          if constexpr (!(COEF_LA || LFO_LA) || !GROOVE_FAST_TABLE_LOOP || !(LFO_MODE == LFO_F32 || (GROOVE_FAST_TABLE_LOOP >= 2 && CL != OSC_ANY))) return false;
          else return __builtin_amdgcn_readfirstlane((int)((!COEF_LA || fu.tab != 0u) && (!LFO_LA || fu.ltab != 0u))) != 0;
        },
        [&](uint32_t k, float& L, float& R) {
          constexpr int TABS = (COEF_LA ? 1 : 0) | (LFO_LA ? 2 : 0);
          double mod = 0.0;
          float tlfo = 0.0f, tamp = 0.0f;
          if constexpr (COEF_LA) {
            sc.coef_f = CoefTab::load<Lp24CoefF>(k & (CoefTab::kFrames - 1), Lay::kStride);
            if constexpr (Lay::kAmp) tamp = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kAmpOff);
          }
          if constexpr (LFO_LA) {
            if constexpr (LFO_MODE == LFO_F64_SMOOTH) mod = CoefTab::load<double>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff);
            else { if (!COEF_LA || (p.flags & WF_LFO_AMP)) tlfo = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); }
          }
          welsh_frame<false, RETUNE, LFO_MODE, C1, C2, CL, true, REST, true, true, Lay::kAmp, TABS>(p, s, rc, sc, L, R, 1u, 1u, mod, tlfo, tamp);
        },
        [&](uint32_t seg, bool live) {
          welsh_segment_end_hoisted<CL == LFO_UNUSED>(p, s, seg, live);
          if constexpr (COEF_LA) { if (fu.tab != 0u) { if (live) s.fil.value = env_last_value_of(s.fil); sc.prev_pct = __builtin_nanf(""); } }
          if constexpr (LFO_LA) { if (fu.ltab != 0u && live) { s.lfo.phase += (uint64_t)seg * p.lfo_inc; if constexpr (LFO_MODE == LFO_F64_SMOOTH) welsh_lfo_reseed_smooth<CL>(p, s, sc); } }
        },
        [&](uint32_t f, uint32_t mine) { welsh_diag_zero(dw, s, active, f, mine); });
    welsh_scratch_f32_end(s, sc);
    return;
  }
  if (UNIFORM && !RETUNE) sc.coef = make_scalar(sc.coef);
  if constexpr (UNIFORM) {
    constexpr bool COEF_LA = RETUNE && LFO_MODE != LFO_F64 && GROOVE_COEF_LOOKAHEAD; // (the exact-f64 kind keeps its own coefficient forms: resonance routing, lp24_coefd_from_fc)
    constexpr bool LFO_LA = LFO_MODE != LFO_F64 && CL != LFO_UNUSED && GROOVE_LFO_LOOKAHEAD;
    typedef TabLayout<false, COEF_LA, LFO_LA, LFO_MODE == LFO_F64_SMOOTH> Lay;
    WaveUniform fu{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u, 0u, 0u};
    run_frames_segmented<FUSED, FASTONLY && (COEF_LA || LFO_LA)>(
        frames, n, v, active, ch_stride, out, rows, prow,
        [&](float& L, float& R) { welsh_frame<true, RETUNE, LFO_MODE, C1, C2, CL, false, REST>(p, s, rc, sc, L, R); },
        [&](bool& live) { const uint32_t k = welsh_segment_begin(p, s, live); welsh_segment_start_hoisted(s, sc); return k; },
        [&](bool live, uint32_t seg) {
          fu.tab = 0u; fu.ltab = 0u;
          if constexpr (COEF_LA || LFO_LA) { if (FASTONLY || seg >= CoefTab::kMinSegment) fu = wave_uniform<COEF_LA, LFO_LA, LFO_MODE, CL, Lay::kAmp>(p, s, sc, live, rc.look); }
          if constexpr (FASTONLY && (COEF_LA || LFO_LA)) { // the promise this copy was chosen on (welsh_wave_tables_up): counted if it does not hold
            if (!((!COEF_LA || fu.tab != 0u) && (!LFO_LA || fu.ltab != 0u)) && __ballot(live) != 0 && (threadIdx.x & 63u) == 0) diag_count_fast_table_miss(dw.diag);
          }
        },
        [&](uint32_t k) { if constexpr (COEF_LA || LFO_LA) { if ((fu.tab | fu.ltab) != 0u && (k & (CoefTab::kFrames - 1)) == 0) wave_tab_fill<false, COEF_LA, LFO_LA, LFO_MODE, CL>(p, rc, fu, k); } },
        [&](uint32_t k, float& L, float& R) {
          uint32_t tab = 0u, ltab = 0u;
          double mod = 0.0;
          float tamp = 0.0f;
          if constexpr (COEF_LA) {
            tab = fu.tab;
            if (tab != 0u) {
              sc.coef = CoefTab::load<Lp24CoefD>(k & (CoefTab::kFrames - 1), Lay::kStride);
              if constexpr (Lay::kAmp) tamp = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kAmpOff);
            }
          }
          float tlfo = 0.0f;
          if constexpr (LFO_LA) {
            ltab = fu.ltab;
            if constexpr (LFO_MODE == LFO_F64_SMOOTH) { if (ltab != 0u) mod = CoefTab::load<double>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); }
            else { if (ltab != 0u && ((p.flags & WF_LFO_AMP) || tab == 0u)) tlfo = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); } // (a cutoff-only LFO is in the coefficients already)
          }
          welsh_frame<false, RETUNE, LFO_MODE, C1, C2, CL, true, REST, true, false, Lay::kAmp>(p, s, rc, sc, L, R, tab, ltab, mod, tlfo, tamp);
        },
        [&]() { // (the f64-filter bodies keep one loop: see the fp32-filter copy above)
          if constexpr (!(COEF_LA || LFO_LA) || GROOVE_FAST_TABLE_LOOP < 3 || LFO_MODE != LFO_F32) return false;
          else return __builtin_amdgcn_readfirstlane((int)((!COEF_LA || fu.tab != 0u) && (!LFO_LA || fu.ltab != 0u))) != 0;
        },
        [&](uint32_t k, float& L, float& R) {
          constexpr int TABS = (COEF_LA ? 1 : 0) | (LFO_LA ? 2 : 0);
          double mod = 0.0;
          float tlfo = 0.0f, tamp = 0.0f;
          if constexpr (COEF_LA) {
            sc.coef = CoefTab::load<Lp24CoefD>(k & (CoefTab::kFrames - 1), Lay::kStride);
            if constexpr (Lay::kAmp) tamp = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kAmpOff);
          }
          if constexpr (LFO_LA) {
            if constexpr (LFO_MODE == LFO_F64_SMOOTH) mod = CoefTab::load<double>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff);
            else { if (!COEF_LA || (p.flags & WF_LFO_AMP)) tlfo = CoefTab::load<float>(k & (CoefTab::kFrames - 1), Lay::kStride, Lay::kModOff); }
          }
          welsh_frame<false, RETUNE, LFO_MODE, C1, C2, CL, true, REST, true, false, Lay::kAmp, TABS>(p, s, rc, sc, L, R, 1u, 1u, mod, tlfo, tamp);
        },
        [&](uint32_t seg, bool live) {
          welsh_segment_end_hoisted<CL == LFO_UNUSED>(p, s, seg, live);
          if constexpr (COEF_LA) { if (fu.tab != 0u) { if (live) s.fil.value = env_last_value_of(s.fil); sc.prev_pct = __builtin_nanf(""); } }
          if constexpr (LFO_LA) { if (fu.ltab != 0u && live) { s.lfo.phase += (uint64_t)seg * p.lfo_inc; if constexpr (LFO_MODE == LFO_F64_SMOOTH) welsh_lfo_reseed_smooth<CL>(p, s, sc); } }
        },
        [&](uint32_t f, uint32_t mine) { welsh_diag_zero(dw, s, active, f, mine); });
  } else {
    run_frames<FUSED>(frames, n, v, active, ch_stride, out, rows, prow, [&](uint32_t f, float& L, float& R) {
      if (f == 0) welsh_frame<true, RETUNE, LFO_MODE, C1, C2, CL>(p, s, rc, sc, L, R);
      else welsh_frame<false, RETUNE, LFO_MODE, C1, C2, CL>(p, s, rc, sc, L, R);
    });
  }
}
// Generic form: per-lane parameters (any mix of patches inside a wave; exec-masked branches).
// Per-lane form: any mix of patches inside a wave (exec-masked branches).  Used for banks
// whose patches are interleaved lane by lane.
template <bool FUSED>
__global__ __launch_bounds__(kThreads) void welsh_render_kernel(
    const uint32_t* __restrict__ params, uint32_t* __restrict__ state, uint32_t n, uint32_t frames,
    size_t ch_stride, float* __restrict__ out, float* __restrict__ rows, RenderConsts rc) {
  const uint32_t v0 = blockIdx.x * kThreads + threadIdx.x;
  const bool active = v0 < n;
  const uint32_t v = active ? v0 : n - 1; // tail lanes shadow the last voice and store nothing
  const WelshParams p = soa_load<WelshParams>(params, n, v);
  WelshState s = soa_load<WelshState>(state, n, v);
  welsh_block<FUSED, true>(p, s, rc, frames, n, v, active, ch_stride, out, rows, blockIdx.x);
  if (active) soa_store(state, n, v, s);
}
// Wave-uniform form.  The host cuts the bank into VIRTUAL WAVES: maximal runs of consecutive
// voices that share one patch, at most 64 long (a 64-lane group that straddles two patches
// becomes two partly filled waves).  Each wavefront of the launch takes one descriptor with
// scalar loads — patch parameters land in SGPRs, waveform / routing dispatch is scalar
// branching — and works on voices [vbase, vbase + count).
//
// Workgroup KINDS.  What a patch needs decides which instantiation (code + register budget) its
// workgroup runs in:
//   - the LFO mode (dsp_core.h: none in f64 / smooth recurrences / exact f64) and whether the
//     filter is retuned per frame (envelope- or LFO-driven cutoff: per-lane f64 coefficients, the
//     tan / reciprocal path) or static (coefficients in SGPRs): six BASE KINDS, one kernel each,
//     launched concurrently on forked streams over their slices of a kind-sorted workgroup list.
//     Separate kernels because register allocation is per kernel: the cheapest base kind fits 72
//     VGPRs (7 waves / SIMD), the most expensive needs 133 (3 waves); inside one kernel everybody
//     paid for the maximum.
//   - the CLASSES of the LFO and the two audio oscillators (dsp_core.h, "Oscillator CLASSES"): inside
//     every base kind's kernel (fused and materialised form; the two exact-f64 kinds since round 6), one
//     scalar switch per workgroup calls the copy of the whole block body compiled for its class triple
//     (6 x 5 x 5 copies; 3 x 5 x 5 for the smooth-LFO kinds).  The copies are NOT inlined:
//     each is a function with its own register allocation (inlined, their hoisted loop invariants
//     all became live across the switch and every copy spilled in its hot loop), and all copies of
//     a base kind need about the same registers, so the kernel's budget fits them all.  (One kernel
//     per class pair was tried too: 32 small launches per block do not run concurrently - the
//     hardware queues are few - and the block took 1.5x as long.)
// The host builds every workgroup from waves that need the SAME kind (welsh_upload_params: the wave
// descriptors are ordered by kind, a kind's last workgroup is filled up with empty waves), so no
// wave runs in a more demanding instantiation than its patch asks for.
struct WaveDesc {
  WelshParams p;
  uint32_t vbase, count;
};
constexpr int kBaseKinds = 6;                                     // LFO mode x retune
constexpr int kClassCombos = LFO_CLASSES * OSC_CLASSES * OSC_CLASSES;  // (LFO class, oscillator 1 class, oscillator 2 class)
constexpr int kWgKinds = kBaseKinds * kClassCombos;                    // sort key of the workgroup list
__host__ __device__ constexpr int wg_base_kind_of(int lfo_mode, bool retune) {
  // cost order (cheap to expensive): F32 static, F32 retune, SMOOTH static, SMOOTH retune, F64 static, F64 retune
  return (lfo_mode == LFO_F32 ? 0 : (lfo_mode == LFO_F64_SMOOTH ? 2 : 4)) + (retune ? 1 : 0);
}
__host__ __device__ constexpr int wg_class_combo(int cl, int c1, int c2) { return (cl * OSC_CLASSES + c1) * OSC_CLASSES + c2; }
__host__ __device__ constexpr int wg_kind_of(int base_kind, int cl, int c1, int c2) { return base_kind * kClassCombos + wg_class_combo(cl, c1, c2); }
__host__ __device__ constexpr bool wg_base_kind_specialised(int base_kind) { (void)base_kind; return true; } // (== dsp_core.h welsh_base_kind_specialised; round 6: all six)
template <int LFO_MODE, bool RETUNE> struct WavesBudget;
template <> struct WavesBudget<LFO_F32, false> { static constexpr int value = GROOVE_WAVES_F32_STATIC; };
template <> struct WavesBudget<LFO_F32, true> { static constexpr int value = GROOVE_WAVES_F32_RETUNE; };
template <> struct WavesBudget<LFO_F64_SMOOTH, false> { static constexpr int value = GROOVE_WAVES_SMOOTH_STATIC; };
template <> struct WavesBudget<LFO_F64_SMOOTH, true> { static constexpr int value = GROOVE_WAVES_SMOOTH_RETUNE; };
template <> struct WavesBudget<LFO_F64, false> { static constexpr int value = GROOVE_WAVES_F64; };
template <> struct WavesBudget<LFO_F64, true> { static constexpr int value = GROOVE_WAVES_F64; };
// The one kernel argument (so that the non-inlined bodies can read it from the kernarg segment
// with scalar loads instead of taking a dozen uniform values through VGPR arguments).
struct UniformArgs {
  const WaveDesc* waves; uint32_t* state; float* out; float* rows; const uint32_t* wg_list; const uint8_t* wg_cls;
  const uint8_t* wg_f32; // per workgroup of the list: its patches carry WF_FILTER_F32 (read by the fused per-kind kernels only)
  size_t ch_stride; RenderConsts rc; uint32_t n_waves, n, frames, n_wgs; // out: the planar block (block-writing form); rows: partial[workgroup][ch][frame] (both forms)
  TpPrev prev; // groove_bank_render_mix_deferred: the previous block's rows, to be put on their bus by this launch (the all-kinds and role-split kernels)
  uint32_t* diag = nullptr; // the context's DiagCounters (diag.h)
  unsigned long long* heartbeat = nullptr; // -DGROOVE_HEARTBEAT builds only (diag.h)
};
typedef const __attribute__((address_space(4))) UniformArgs* UniformArgsPtr; // kernarg segment: scalar loads
__device__ __forceinline__ UniformArgsPtr uniform_args_scalar(UniformArgsPtr a) { // arguments of a call travel in VGPRs
  const uint64_t bits = (uint64_t)a;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
  return (UniformArgsPtr)(((uint64_t)hi << 32) | lo);
}
template <bool FUSED, int LFO_MODE, bool RETUNE, int C1, int C2, int CL, bool REST = false, bool F32OK = false, bool FAST = false>
__device__ __forceinline__ void welsh_uniform_body_impl(UniformArgsPtr a) {
  const uint32_t wg = a->wg_list[GROOVE_WG_SLOT(a->n_wgs)]; // scalar load: the workgroup of virtual waves this block renders
  const uint32_t n_waves = a->n_waves, n = a->n;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w0 = wg * kWaves + (threadIdx.x >> 6);
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(w0, n_waves - 1));
  const WaveDesc d = make_scalar(a->waves[w]);
  const bool active = (w0 < n_waves) && (lane < d.count);
  const uint32_t v = active ? d.vbase + lane : d.vbase; // idle lanes shadow the run's first voice
  WelshState s = soa_load<WelshState>(a->state, n, v);
  RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi}; rc.look = a->rc.look;
  // pinned in VGPRs for the block: as literals / SGPRs each costs a v_mov on every retuning frame (the instructions
  // that use them take one constant-bus operand): +2.5 % in the all-voices window of the million-voice project
  if constexpr (RETUNE) asm volatile("" : "+v"(rc.tan_k1), "+v"(rc.tan_k2), "+v"(rc.log2_x0), "+v"(rc.x_hi));
  welsh_block<FUSED, RETUNE, LFO_MODE, true, C1, C2, CL, REST, F32OK, FAST>(d.p, s, rc, a->frames, n, v, active, a->ch_stride, a->out, a->rows, wg, DiagWhere{a->diag, wg, w, d.count});
  if (active) soa_store(a->state, n, v, s);
}
// Internal linkage + no `tail` marker on the kernels' calls: the compiler's inter-procedural register allocation then
// knows every caller of a body, the body saves and restores NO callee-saved registers (TargetFrameLowering's no-CSR
// path wants a local, non-recursive function none of whose call sites is a tail call), and the kernel — which keeps
// nothing alive across the call — pays nothing either.  As linkonce_odr functions every body spilled 24-38 VGPRs to
// scratch in its prologue and read them back at its end: 84-172 bytes of scratch per lane, ~0.2-0.3 GB per
// million-voice block of HBM traffic that served nothing (round 2's PMC passes).
#define GROOVE_NO_TAIL_CALLS __attribute__((disable_tail_calls))
#define GROOVE_BODY_LINKAGE static
template <bool FUSED, int LFO_MODE, bool RETUNE, int C1, int C2, int CL, bool F32OK = false, bool FAST = false>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_uniform_body(UniformArgsPtr a) {
  // the class bodies only ever see waves of their own classes (dsp_core.h, REST)
  welsh_uniform_body_impl<FUSED, LFO_MODE, RETUNE, C1, C2, CL, true, F32OK, FAST>(uniform_args_scalar(a));
}
// Will this wave's look-ahead tables be up for the whole block?  Note events land between blocks, so lanes whose filter envelope record,
// LFO phase and first-tick flag agree bit for bit when a block starts evolve identically through it: every segment of the block will find
// them in agreement (wave_uniform), whatever its length.  Such a wave takes the FAST copy of its body: the table frames' loop alone in
// the function, no per-lane path beside it.  (Lanes that are idle when the block starts stay idle; a lane that goes idle on the way
// drops out of the comparison.)
__device__ __forceinline__ bool welsh_wave_tables_up(const UniformArgs& a) {
  const uint32_t wg = a.wg_list[GROOVE_WG_SLOT(a.n_wgs)];
  const uint32_t w0 = wg * kWaves + (threadIdx.x >> 6);
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(w0, a.n_waves - 1));
  const uint32_t vbase = a.waves[w].vbase, count = a.waves[w].count, flags = a.waves[w].p.flags;
  if ((a.rc.look & 3u) != 3u) return false;
  if (((flags >> WF_LFO_WAVE_SHIFT) & 15u) == (uint32_t)GROOVE_WAVE_NOISE) return false;
  const bool active = (w0 < a.n_waves) && ((threadIdx.x & 63u) < count);
  const uint32_t v = active ? vbase + (threadIdx.x & 63u) : vbase;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, (int)(sizeof(WelshState) / 4 * a.n * 4u), 0x00020000);
  auto word = [&](uint32_t k) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(v * 4u), (int)(k * a.n * 4u), 0); };
  constexpr uint32_t kAmp = offsetof(WelshState, amp) / 4, kFil = offsetof(WelshState, fil) / 4, kLfo = offsetof(WelshState, lfo) / 4, kFlags = offsetof(WelshState, vflags) / 4;
  const bool live = active && word(kAmp + offsetof(EnvState, state) / 4) != ENV_IDLE;
  const uint64_t mask = __ballot(live);
  if (mask == 0) return true; // (nothing sounds: no frame runs)
  const int l0 = __builtin_ctzll(mask);
  bool same = true;
#pragma unroll
  for (uint32_t k = 0; k < sizeof(EnvState) / 4 + 3; ++k) {
    const uint32_t x = word(k < sizeof(EnvState) / 4 ? kFil + k : k == sizeof(EnvState) / 4 ? kLfo : k == sizeof(EnvState) / 4 + 1 ? kLfo + 1 : kFlags);
    same = same && x == (uint32_t)__builtin_amdgcn_readlane((int)x, l0);
  }
  return __ballot(live && !same) == 0;
}
// A workgroup whose voices are all silent with both envelopes idle (unused polyphony, voices past
// their release) contributes zeros and changes nothing but idle-plateau counters: it writes its zero
// rows and leaves.  Two state words per lane, one vote; nothing else is loaded.  Returns true if
// the workgroup is done.
__device__ __forceinline__ bool welsh_idle_workgroup(const UniformArgs& a) {
  const uint32_t wg = a.wg_list[GROOVE_WG_SLOT(a.n_wgs)];
  const uint32_t w0 = wg * kWaves + (threadIdx.x >> 6);
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(w0, a.n_waves - 1));
  const uint32_t vbase = a.waves[w].vbase, count = a.waves[w].count;
  const bool active = (w0 < a.n_waves) && ((threadIdx.x & 63u) < count);
  const uint32_t v = active ? vbase + (threadIdx.x & 63u) : vbase;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.state, 0, (int)(sizeof(WelshState) / 4 * a.n * 4u), 0x00020000);
  constexpr uint32_t kAmpWord = offsetof(WelshState, amp) / 4 + offsetof(EnvState, state) / 4;
  constexpr uint32_t kFilWord = offsetof(WelshState, fil) / 4 + offsetof(EnvState, state) / 4;
  const uint32_t sa = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(v * 4u), (int)(kAmpWord * a.n * 4u), 0);
  const uint32_t sf = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(v * 4u), (int)(kFilWord * a.n * 4u), 0);
  __shared__ int busy_waves;
  if (threadIdx.x == 0) busy_waves = 0;
  __syncthreads();
  if (!__all(!active || (sa == ENV_IDLE && sf == ENV_IDLE)) && (threadIdx.x & 63u) == 0) atomicAdd(&busy_waves, 1);
  __syncthreads();
  if (busy_waves != 0) return false;
  float* __restrict__ rows = a.rows + (size_t)wg * 2 * a.frames; // partial[wg][ch][frame]
  for (uint32_t t = threadIdx.x; t < 2 * a.frames; t += kThreads) rows[t] = 0.0f;
  return true;
}
// One scalar switch on the workgroup's class combination, to the block body compiled for it.
template <bool FUSED, int LFO_MODE, bool RETUNE, bool F32OK = false, bool FAST = false>
__device__ __forceinline__ void welsh_dispatch_class(uint32_t cls, UniformArgsPtr ka) {
  // (a FAST copy exists where the kind has a table at all: not for the static F32 kind's LFO-less bodies)
#define GROOVE_CLS_CASE(CL, C1, C2) case wg_class_combo(CL, C1, C2): welsh_uniform_body<FUSED, LFO_MODE, RETUNE, C1, C2, CL, F32OK, (FAST && (RETUNE || CL != LFO_UNUSED))>(ka); break;
#define GROOVE_CLS_ROW(CL, C1) GROOVE_CLS_CASE(CL, C1, 0) GROOVE_CLS_CASE(CL, C1, 1) GROOVE_CLS_CASE(CL, C1, 2) GROOVE_CLS_CASE(CL, C1, 3) GROOVE_CLS_CASE(CL, C1, 4)
#define GROOVE_CLS_PLANE(CL) GROOVE_CLS_ROW(CL, 0) GROOVE_CLS_ROW(CL, 1) GROOVE_CLS_ROW(CL, 2) GROOVE_CLS_ROW(CL, 3) GROOVE_CLS_ROW(CL, 4)
  switch (cls) {
    GROOVE_CLS_PLANE(OSC_ANY) GROOVE_CLS_PLANE(OSC_TRIANGLE) GROOVE_CLS_PLANE(OSC_SINE)
    default:
      if constexpr (LFO_MODE != LFO_F64_SMOOTH) { // (the smooth kinds carry the sine, triangle and `any` LFO copies only: welsh_body_classes)
        switch (cls) {
          GROOVE_CLS_PLANE(OSC_PULSE) GROOVE_CLS_PLANE(OSC_SAW) GROOVE_CLS_PLANE(LFO_UNUSED)
          default: break;
        }
      }
      break;
  }
#undef GROOVE_CLS_PLANE
#undef GROOVE_CLS_ROW
#undef GROOVE_CLS_CASE
}
// SPECIALISED = false: the whole launch runs the OSC_ANY x OSC_ANY body (wg_cls is not read).
template <bool FUSED, int LFO_MODE, bool RETUNE, bool SPECIALISED>
__global__ __launch_bounds__(kThreads, (WavesBudget<LFO_MODE, RETUNE>::value)) GROOVE_NO_TAIL_CALLS void welsh_render_uniform_kernel(UniformArgs a) {
  // (No s_setprio for the class-specialised kinds: raising the f64-LFO kinds' wave priority paid when a block's kernels were forked and
  // joined; with the blocks pipelined the longest kernel is the most numerous kind, and any priority
  // costs 5 % — measured: none 0.460 ms, f64-LFO kinds raised 0.484, F32-retune raised 0.513.)
  const UniformArgsPtr ka = (UniformArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(); // == &a, in the constant address space
  // (Round 6, the exact-f64 kinds beside the mix kernel's three launches: their few wavefronts — 1.9 % of a library-proportioned bank's
  // voices — take the block's longest walk, ~410 us beside mix launches of ~350, 227 us alone.  s_setprio(3) for these kernels changed
  // nothing: 0.4067 against 0.4052 ms per block; a wave's walk is a serial instruction stream 2.3x the average voice's, and no priority
  // shortens it.  Left out.)
#ifdef GROOVE_HEARTBEAT /* diag.h: workgroups started / finished, counted in host memory the host can read while the device is stuck */
  unsigned long long* hb = a.heartbeat;
  if (hb && threadIdx.x == 0) __hip_atomic_fetch_add(hb + 0, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#define GROOVE_HB_DONE do { __syncthreads(); if (hb && threadIdx.x == 0) __hip_atomic_fetch_add(hb + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define GROOVE_HB_DONE do { } while (0)
#endif
  if constexpr (FUSED) { if (welsh_idle_workgroup(a)) { GROOVE_HB_DONE; return; } }
  if constexpr (!SPECIALISED) {
    welsh_uniform_body_impl<FUSED, LFO_MODE, RETUNE, OSC_ANY, OSC_ANY, OSC_ANY>(ka);
  } else {
    // The FUSED per-kind kernels honour WF_FILTER_F32 (a second set of block bodies); the block-writing per-kind kernels and the
    // all-kinds, role-split and time-parallel kernels keep the f64 filter for every voice.
    const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wg_cls[GROOVE_WG_SLOT(a.n_wgs)]);
    if constexpr (FUSED && LFO_MODE != LFO_F64) {
      if (__builtin_amdgcn_readfirstlane((int)a.wg_f32[GROOVE_WG_SLOT(a.n_wgs)])) welsh_dispatch_class<FUSED, LFO_MODE, RETUNE, true>(cls, ka);
      else welsh_dispatch_class<FUSED, LFO_MODE, RETUNE, false>(cls, ka);
    } else {
      welsh_dispatch_class<FUSED, LFO_MODE, RETUNE, false>(cls, ka);
    }
  }
  GROOVE_HB_DONE;
#undef GROOVE_HB_DONE
}
// All (class-specialised) base kinds in ONE launch, for banks too small to fill the machine: there a block is
// bound by one wavefront's serial walk of its frames, register budgets do not matter (the kernel takes
// the largest), and what counts is that every workgroup starts at once instead of queueing behind
// the few hardware queues that several per-kind launches share.  wg_base[] = base kind per workgroup.
#ifdef GROOVE_WELSH_ANY_TU // defined by the two translation units that own this kernel (fused form, block-writing form)
template <bool FUSED>
__global__ __launch_bounds__(kThreads, GROOVE_WAVES_ANY) GROOVE_NO_TAIL_CALLS void welsh_render_uniform_any_kernel(UniformArgs a, const uint8_t* __restrict__ wg_base) {
  const UniformArgsPtr ka = (UniformArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (FUSED) { tp_reduce_prev(a.prev, threadIdx.x, blockIdx.x, gridDim.x); if (welsh_idle_workgroup(a)) return; }
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg_base[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wg_cls[GROOVE_WG_SLOT(a.n_wgs)]);
  // (FUSED: the FAST copies of the bodies, as in the mix kernel below — here, where a bank does not fill the chip and a block is one
  // wavefront's walk of its frames, a shorter frame is a shorter block)
  bool fastw = false;
  if constexpr (FUSED) {
    fastw = __builtin_amdgcn_readfirstlane((int)welsh_wave_tables_up(a)) != 0;
    if (fastw && (a.rc.look & 4u) && (threadIdx.x & 63u) == 0) diag_count_fast_wave(a.diag);
  }
#define GROOVE_ANY_CASE(MODE, RETUNE)                                                          \
  if constexpr (FUSED) {                                                                       \
    if (fastw) welsh_dispatch_class<FUSED, MODE, RETUNE, false, true>(cls, ka);                \
    else welsh_dispatch_class<FUSED, MODE, RETUNE>(cls, ka);                                   \
  } else {                                                                                     \
    welsh_dispatch_class<FUSED, MODE, RETUNE>(cls, ka);                                        \
  }
  switch (base) {
    case wg_base_kind_of(LFO_F32, false): GROOVE_ANY_CASE(LFO_F32, false) break;
    case wg_base_kind_of(LFO_F32, true): GROOVE_ANY_CASE(LFO_F32, true) break;
    case wg_base_kind_of(LFO_F64_SMOOTH, false): GROOVE_ANY_CASE(LFO_F64_SMOOTH, false) break;
    default: GROOVE_ANY_CASE(LFO_F64_SMOOTH, true) break;
    // (the exact-f64 base kinds 4 and 5 are never in this launch: groove_hip.hip launch_small_uniform)
  }
#undef GROOVE_ANY_CASE
}
#endif // GROOVE_WELSH_ANY_TU
// The four class-specialised base kinds in ONE kernel at the per-kind kernels' own register budget (all four are budgeted for five
// waves per SIMD since round 5), the fp32-filter bodies included: the MIX kernel of big banks (round 6).  A block of a big bank is
// THREE launches of it, one per kind stream, each over a third of the workgroup list taken with a stride of three (groove_hip.hip
// welsh_upload_params: every launch carries the bank's own mix of kinds, most expensive first), instead of one launch per base kind:
// with per-kind launches a stream's time per block is what ITS kinds cost — 460 / 455 / 222 us on the three streams for the 32-patch
// benchmark table (profiles/r05_welsh-1m-window_summary.json), 462 / 546 / 266 for the library-proportioned one — and the slowest stream is
// the step; thirds of everything are balanced whatever the patches are, and every stream has one launch per block (a launch cannot
// take less than one wavefront's walk of the block, ~180 us: two launches on a stream are two such floors).
// FAST copies (end of round 6).  Every body with a look-ahead table exists twice in this kernel: the copy above, whose segments ask at
// their start whether the tables are up and run the per-lane frame when they are not, and a FAST copy that holds the table frames' loop
// and nothing else (welsh_block's FASTONLY) — chosen per WAVE when the block starts, by welsh_wave_tables_up: note events land between
// blocks, so a wave whose live voices agree on the filter envelope's record, the LFO's phase and the first-tick flag when a block starts
// will find them in agreement at every segment of the block.  With one hot loop per function the compiler keeps it in registers in every
// body (the same loop beside the per-lane loop cost 2 - 20 scratch accesses a frame in the f64-filter and smooth-f64 bodies:
// GROOVE_FAST_TABLE_LOOP): 57 - 110 instructions a frame where the shared loop's table path ran 130 - 170.  In one job, headline
// 0.3301 - 0.3331 -> 0.3118 - 0.3172 ms per block, the library-proportioned bank 0.3376 -> 0.3293, the whole timeline 0.2751 -> 0.2628.  The
// promise is a counted assertion (diag.h fast_table_misses, required to be 0 like zero_segments).
#ifndef GROOVE_WAVES_MIX
#define GROOVE_WAVES_MIX 5
#endif
#ifdef GROOVE_WELSH_MIX_TU // defined by the translation units that own this kernel (csrc/welsh_class.hip, -DGROOVE_BASE_KIND=10 fused, 11 block-writing)
template <bool FUSED>
__global__ __launch_bounds__(kThreads, GROOVE_WAVES_MIX) GROOVE_NO_TAIL_CALLS void welsh_render_uniform_mix_kernel(UniformArgs a, const uint8_t* __restrict__ wg_base) {
  const UniformArgsPtr ka = (UniformArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (FUSED) { if (welsh_idle_workgroup(a)) return; }
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg_base[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wg_cls[GROOVE_WG_SLOT(a.n_wgs)]);
  bool f32 = false;
  if constexpr (FUSED) f32 = __builtin_amdgcn_readfirstlane((int)a.wg_f32[GROOVE_WG_SLOT(a.n_wgs)]) != 0;
  bool fastw = false;
  if constexpr (FUSED) {
    fastw = __builtin_amdgcn_readfirstlane((int)welsh_wave_tables_up(a)) != 0;
    if (fastw && (a.rc.look & 4u) && (threadIdx.x & 63u) == 0) diag_count_fast_wave(a.diag); // (tests only: groove_set_look_ahead(7))
  }
#define GROOVE_MIX_CASE(MODE, RETUNE)                                                          \
  if constexpr (FUSED) {                                                                       \
    if (fastw) { if (f32) welsh_dispatch_class<FUSED, MODE, RETUNE, true, true>(cls, ka); else welsh_dispatch_class<FUSED, MODE, RETUNE, false, true>(cls, ka); } \
    else if (f32) welsh_dispatch_class<FUSED, MODE, RETUNE, true>(cls, ka);                         \
    else welsh_dispatch_class<FUSED, MODE, RETUNE, false>(cls, ka);                            \
  } else {                                                                                     \
    welsh_dispatch_class<FUSED, MODE, RETUNE, false>(cls, ka);                                 \
  }
  switch (base) {
    case wg_base_kind_of(LFO_F32, false): GROOVE_MIX_CASE(LFO_F32, false) break;
    case wg_base_kind_of(LFO_F32, true): GROOVE_MIX_CASE(LFO_F32, true) break;
    case wg_base_kind_of(LFO_F64_SMOOTH, false): GROOVE_MIX_CASE(LFO_F64_SMOOTH, false) break;
    default: GROOVE_MIX_CASE(LFO_F64_SMOOTH, true) break;
    // (the exact-f64 base kinds 4 and 5 are never in this launch: their own kernels, on their own stream)
  }
#undef GROOVE_MIX_CASE
}
#endif // GROOVE_WELSH_MIX_TU
static_assert(OSC_CLASSES == 5 && LFO_CLASSES == 6 && kClassCombos <= 256, "the class switch above lists 6 x 5 x 5 combinations, one byte each");
// Launchers of the four class-specialised fused kernels, one translation unit each
// (csrc/welsh_class.hip, -DGROOVE_BASE_KIND=0..3) so that they compile in parallel.
// `done` (optional, every launcher below): an event that completes WITH the kernel — bound to the dispatch's own completion signal
// (hipExtLaunchKernelGGL) instead of recorded behind it: a recorded event is a barrier packet of its own, ~5 us of its stream's timeline
// before the next kernel of the stream starts.
template <class K, class... Args>
static inline void launch_bound(K kernel, dim3 grid, dim3 blk, hipStream_t st, hipEvent_t done, Args... args) {
  if (done) hipExtLaunchKernelGGL(kernel, grid, blk, 0, st, nullptr, done, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, blk, 0, st, args...);
}
void launch_welsh_uniform_specialised_0(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr);
void launch_welsh_uniform_specialised_1(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr);
void launch_welsh_uniform_specialised_2(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr);
void launch_welsh_uniform_specialised_3(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr);
void launch_welsh_uniform_specialised_4(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr); // exact-f64 LFO, static filter (round 6)
void launch_welsh_uniform_specialised_5(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr); // exact-f64 LFO, retuned filter / resonance routing
void launch_welsh_uniform_any(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done = nullptr); // fused: csrc/welsh_class.hip, -DGROOVE_BASE_KIND=9
void launch_welsh_uniform_any_unfused(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done = nullptr); // writes the voice block: -DGROOVE_BASE_KIND=8
void launch_welsh_uniform_mix(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done = nullptr); // the mix kernel, fused: -DGROOVE_BASE_KIND=10
void launch_welsh_uniform_mix_unfused(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done = nullptr); // ... writing the voice block: -DGROOVE_BASE_KIND=11

template <bool FUSED>
__global__ __launch_bounds__(kThreads) void fm_render_kernel(
    const uint32_t* __restrict__ params, uint32_t* __restrict__ state, uint32_t n, uint32_t frames,
    size_t ch_stride, float* __restrict__ out, float* __restrict__ rows) {
  const uint32_t v0 = blockIdx.x * kThreads + threadIdx.x;
  const bool active = v0 < n;
  const uint32_t v = active ? v0 : n - 1;
  const FmParams p = soa_load<FmParams>(params, n, v);
  FmState s = soa_load<FmState>(state, n, v);
  run_frames<FUSED>(frames, n, v, active, ch_stride, out, rows, blockIdx.x, [&](uint32_t f, float& L, float& R) {
    if (f == 0) fm_frame<true>(p, s, L, R);
    else fm_frame<false>(p, s, L, R);
  });
  if (active) soa_store(state, n, v, s);
}

// a7 SamplerVoice: pointer stepping; the shared bank is a gather served from L2 / MALL.
template <bool FUSED>
__global__ __launch_bounds__(kThreads) void sampler_render_kernel(
    const uint32_t* __restrict__ params, uint32_t* __restrict__ state, uint32_t n, uint32_t frames,
    size_t ch_stride, float* __restrict__ out, float* __restrict__ rows, const float* __restrict__ bank) {
  const uint32_t v0 = blockIdx.x * kThreads + threadIdx.x;
  const bool active = v0 < n;
  const uint32_t v = active ? v0 : n - 1;
  const SamplerParams p = soa_load<SamplerParams>(params, n, v);
  SamplerState s = soa_load<SamplerState>(state, n, v);
  // chunks of 16 frames: sixteen fetches in flight instead of one (sampler_chunk); mono duplicated to both channels
  constexpr uint32_t C = 16, K = FusedAcc::kChunk;
  static_assert(C % K == 0, "whole flush periods of the fused epilogue per chunk");
  FusedAcc acc(blockIdx.x);
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t count = min(C, frames - f0);
    float x[C];
    sampler_chunk<C>(p, s, bank, count, x);
#pragma unroll
    for (uint32_t k = 0; k < C; ++k) {
      if (k < count) {
        acc.add(active ? x[k] : 0.0f, active ? x[k] : 0.0f, f0 + k);
        if (!FUSED && active) { out[(size_t)(f0 + k) * n + v] = x[k]; out[ch_stride + (size_t)(f0 + k) * n + v] = x[k]; }
      }
      if ((k & (K - 1)) == K - 1 && k - (K - 1) < count) acc.flush(rows, frames, f0 + k - (K - 1), min(K, count - (k - (K - 1))));
    }
  }
  if (active) soa_store(state, n, v, s);
}

#ifndef GROOVE_WELSH_CLASS_TU // everything below is defined once, in groove_hip.hip
// Fused path, stage 2: column sums of partial[rows][cols] (cols = 2*frames) over row segments.
// (Both stages in one launch — the last workgroup of a column group to finish adds the segments up — was tried and is
// slower: the agent-scope release every workgroup needs before it counts itself done writes the whole L2 back on this
// multi-XCD part, 18.6 us against 6 + 5 for sampler-16384.)
__global__ __launch_bounds__(kThreads) void partial_rows_kernel(
    const float* __restrict__ partial, uint32_t rows, uint32_t cols, uint32_t rows_per_seg,
    float* __restrict__ seg_out /*[segs][cols]*/, float* __restrict__ bus = nullptr, int accumulate = 0) {
  const uint32_t c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= cols) return;
  const uint32_t r0 = blockIdx.y * rows_per_seg, r1 = min(rows, r0 + rows_per_seg);
  // sixteen independent accumulators: a 64-row segment is four rounds of sixteen loads in flight (with four it was
  // sixteen dependent rounds, 6 us for a bank whose whole reduction is one segment)
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.0f;
  uint32_t r = r0;
  for (; r + 16 <= r1; r += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] += partial[(size_t)(r + k) * cols + c];
  }
  for (; r + 4 <= r1; r += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += partial[(size_t)(r + k) * cols + c];
  }
  for (; r < r1; ++r) a[0] += partial[(size_t)r * cols + c];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] += a[k + 8];
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] += a[k + 4];
  const float t = (a[0] + a[1]) + (a[2] + a[3]);
  if (bus) { // a single segment: this IS the column total (what partial_final_kernel would add up and write)
    const uint32_t frames = cols / 2, ch = c / frames, f = c % frames;
    if (accumulate) bus[2 * f + ch] += t; else bus[2 * f + ch] = t;
  } else {
    seg_out[(size_t)blockIdx.y * cols + c] = t;
  }
}
// Stage 3: bus[f][ch] (+)= sum_seg seg[seg][ch*frames + f].  Eight independent accumulators keep
// eight L2 round trips in flight (a single dependent chain made this 29 us for 123 segments).
__global__ void partial_final_kernel(const float* __restrict__ seg, uint32_t segs, uint32_t frames,
                                     float* __restrict__ bus, int accumulate) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * frames) return;
  const size_t cols = 2 * (size_t)frames;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t s = 0;
  for (; s + 8 <= segs; s += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += seg[(size_t)(s + k) * cols + c];
  }
  for (; s < segs; ++s) a[0] += seg[(size_t)s * cols + c];
  const float t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  const uint32_t ch = c / frames, f = c % frames;
  if (accumulate) bus[2 * f + ch] += t; else bus[2 * f + ch] = t;
}

// Every voice back to the state a freshly created bank has (groove_bank_reset): the initial state is
// the same record for every lane, passed by value.
struct StateWords { uint32_t w[64]; };
static_assert(sizeof(WelshState) <= sizeof(StateWords) && sizeof(FmState) <= sizeof(StateWords) && sizeof(SamplerState) <= sizeof(StateWords), "state record fits the reset argument");
__global__ __launch_bounds__(kThreads) void state_fill_kernel(uint32_t* __restrict__ state, uint32_t n, uint32_t words, StateWords init) {
  const uint32_t v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= n) return;
  for (uint32_t i = 0; i < words; ++i) state[(size_t)i * n + v] = init.w[i];
}

// HandlesMidi: one thread per event (voice != ALL) or one thread per voice (voice == ALL).
struct NoteCtx { double sr; };
__global__ void welsh_events_kernel(const groove_note_event* __restrict__ ev, uint32_t n_ev, int all_event,
                                    const uint32_t* __restrict__ params, uint32_t* __restrict__ state,
                                    const double* __restrict__ cold /*[4][n]*/, uint32_t n, double sr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  groove_note_event e;
  uint32_t v;
  if (all_event >= 0) { if (i >= n) return; e = ev[all_event]; v = i; }
  else { if (i >= n_ev) return; e = ev[i]; v = e.voice; if (v >= n) return; }
  const WelshParams p = soa_load<WelshParams>(params, n, v);
  WelshState s = soa_load<WelshState>(state, n, v);
  welsh_note(p, s, cold[v], cold[(size_t)n + v], cold[(size_t)2 * n + v], cold[(size_t)3 * n + v], sr, e.key, e.on != 0);
  soa_store(state, n, v, s);
}
__global__ void fm_events_kernel(const groove_note_event* __restrict__ ev, uint32_t n_ev, int all_event,
                                 const uint32_t* __restrict__ params, uint32_t* __restrict__ state,
                                 const double* __restrict__ ratio, uint32_t n, double sr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  groove_note_event e;
  uint32_t v;
  if (all_event >= 0) { if (i >= n) return; e = ev[all_event]; v = i; }
  else { if (i >= n_ev) return; e = ev[i]; v = e.voice; if (v >= n) return; }
  const FmParams p = soa_load<FmParams>(params, n, v);
  FmState s = soa_load<FmState>(state, n, v);
  fm_note(p, s, ratio[v], sr, e.key, e.on != 0);
  soa_store(state, n, v, s);
}
__global__ void sampler_events_kernel(const groove_note_event* __restrict__ ev, uint32_t n_ev, int all_event,
                                      const uint32_t* __restrict__ params, uint32_t* __restrict__ state, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  groove_note_event e;
  uint32_t v;
  if (all_event >= 0) { if (i >= n) return; e = ev[all_event]; v = i; }
  else { if (i >= n_ev) return; e = ev[i]; v = e.voice; if (v >= n) return; }
  const SamplerParams p = soa_load<SamplerParams>(params, n, v);
  SamplerState s = soa_load<SamplerState>(state, n, v);
  sampler_note(p, s, e.key, e.on != 0);
  soa_store(state, n, v, s);
}
// ------------------------------------------------------------------ mix bus (a15)
// Stage 1: grid (segments, rows) with rows = 2*frames; each workgroup sums one segment of
// one row of the block (row = one channel of one frame across all lanes) → partial[row][seg].
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}
// direct != nullptr (only with n_seg == 1): the row total goes straight to the bus / one-lane block
// that mix_final_kernel would have copied it to (one launch less on a latency-bound path).
__global__ __launch_bounds__(kThreads) void mix_partial_kernel(
    const float* __restrict__ block, uint32_t n, uint32_t frames, size_t ch_stride, uint32_t seg_len,
    float* __restrict__ partial, uint32_t n_seg, float* __restrict__ direct = nullptr, int accumulate = 0, size_t planar_stride = 0) {
  const uint32_t row = blockIdx.y; // ch*frames + f
  const uint32_t seg = blockIdx.x;
  const uint32_t ch = row / frames, f = row % frames;
  const float* __restrict__ src = block + ch * ch_stride + (size_t)f * n;
  const uint32_t lo = seg * seg_len;
  const uint32_t hi = min(n, lo + seg_len);
  float acc = 0.0f;
  if ((n & 3u) == 0 && (seg_len & 3u) == 0) {
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src);
    for (uint32_t i = lo / 4 + threadIdx.x; i < hi / 4; i += kThreads) {
      const float4 q = s4[i];
      acc += (q.x + q.y) + (q.z + q.w);
    }
  } else {
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kThreads) acc += src[i];
  }
  acc = wave_sum(acc);
  __shared__ float red[kThreads / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) t += red[i];
    if (direct) {
      float* dst = planar_stride ? direct + ch * planar_stride + f : direct + 2 * f + ch;
      if (accumulate) *dst += t; else *dst = t;
    } else {
      partial[(size_t)row * n_seg + seg] = t;
    }
  }
}
// Stage 2: bus[f][ch] (+)= sum_seg partial[row][seg]; one thread per row.  planar_stride = 0
// writes the interleaved bus; otherwise a one-lane block dst[ch * planar_stride + f].
__global__ void mix_final_kernel(const float* __restrict__ partial, uint32_t frames, uint32_t n_seg,
                                 float* __restrict__ bus, int accumulate, size_t planar_stride) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= 2 * frames) return;
  float t = 0.0f;
  for (uint32_t s = 0; s < n_seg; ++s) t += partial[(size_t)row * n_seg + s];
  const uint32_t ch = row / frames, f = row % frames;
  float* dst = planar_stride ? bus + ch * planar_stride + f : bus + 2 * f + ch;
  if (accumulate) *dst += t; else *dst = t;
}
// Lane permutation of a planar block: dst[ch][f][e] = src[ch][f][src_lane[e]].  Used when a bank keeps
// its voices in another lane order than the caller (patch-major regrouping): the render writes its
// own order with coalesced rows and this pass hands the caller's order out, again with coalesced
// writes (the gathered reads of neighbouring workgroups share cache lines).
__global__ __launch_bounds__(kThreads) void block_gather_kernel(float* __restrict__ dst, size_t dst_ch_stride, const float* __restrict__ src,
                                                               size_t src_ch_stride, const uint32_t* __restrict__ src_lane, uint32_t n,
                                                               uint32_t frames) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= n) return;
  const uint32_t i = src_lane[e];
  for (uint32_t f = blockIdx.y; f < frames; f += gridDim.y) {
    dst[(size_t)f * n + e] = src[(size_t)f * n + i];
    dst[dst_ch_stride + (size_t)f * n + e] = src[src_ch_stride + (size_t)f * n + i];
  }
}

// dst (+)= src, element-wise over [2][frames][n] blocks with their own channel strides.
__global__ __launch_bounds__(kThreads) void block_add_kernel(
    float* __restrict__ dst, size_t dst_chs, const float* __restrict__ src, size_t src_chs,
    uint32_t n, uint32_t frames, int accumulate) {
  const size_t per_ch = (size_t)frames * n;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < 2 * per_ch; i += (size_t)gridDim.x * kThreads) {
    const size_t ch = i / per_ch, r = i % per_ch;
    const float x = src[ch * src_chs + r];
    float* d = dst + ch * dst_chs + r;
    *d = accumulate ? *d + x : x;
  }
}
// WAV sink quantisation (helpers.rs:79-91): (x * 32767) as i16, truncating, saturating.
__global__ void bus_to_i16_kernel(const float* __restrict__ bus, size_t count, int16_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float v = bus[i] * 32767.0f;
  int q;
  if (v != v) q = 0;
  else if (v >= 32767.0f) q = 32767;
  else if (v <= -32768.0f) q = -32768;
  else q = (int)v;
  out[i] = (int16_t)q;
}

// ------------------------------------------------------------------ effects
// Element-wise kinds (a8 Gain, a9 Bitcrusher, Limiter, Compressor; the Mixer is the identity and launches nothing): stages
// of fx_run_kernel below, alone or fused with their neighbours in a chain.

// The IIR and delay-line effect kernels keep one (channel, lane) pair per thread and walk the
// block sequentially (feedback), but in CHUNKS of C frames: all the loads of a chunk (inputs and
// delay-line reads) are issued back to back before any arithmetic, so a chunk costs one memory
// round trip instead of C.  Delay-line reads of a chunk may not alias its writes, which holds
// when every line is at least C frames long (the host picks C = 1 otherwise).  With only
// 2 * lanes threads (config #3: 8,192 = 128 waves on 1,024 SIMDs) these kernels are latency
// bound, and the chunking is what sets their speed.

// IIR kinds (a3 BiQuad 12 dB, a4 24 dB low-pass): f64 coefficients and state.
// coef: [5 or 6][n] f64; st: [4][2n] f64.
template <int C>
__global__ __launch_bounds__(kThreads) void fx_biquad_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  BiquadCoefD c{coef[lane], coef[(size_t)n + lane], coef[(size_t)2 * n + lane], coef[(size_t)3 * n + lane], coef[(size_t)4 * n + lane]};
  BiquadStateD s{st[t], st[ln + t], st[2 * ln + t], st[3 * ln + t]};
  const float w = wet[lane];
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  // the next chunk's inputs are requested before this chunk's recurrence runs, so the memory round
  // trip of chunk k+1 hides behind the arithmetic of chunk k
  float xn[C];
#pragma unroll
  for (int j = 0; j < C; ++j) xn[j] = (uint32_t)j < frames ? ptr[(size_t)j * n] : 0.0f;
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C];
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = xn[j];
#pragma unroll
    for (int j = 0; j < C; ++j) xn[j] = f0 + C + j < frames ? ptr[(size_t)(f0 + C + j) * n] : 0.0f;
    if (c_n == (uint32_t)C) {
      // A whole chunk, straight-line.  biquad_step's five operations in their order, but scheduled in two sweeps: the
      // feed-forward part b0 x + b1 x1 + b2 x2 of every frame first (independent of each other: they issue back to
      // back), then the feedback part - a1 y1 - a2 y2, the only dependent chain, two operations per frame instead of
      // five.  (Measured: the walk of a bank that does not fill the chip stays at ~13 us of kernel time for 256 frames —
      // 9 us of it the two-operation f64 chain itself, 4 us the stores, which share the loads' counter.)
      double ff[C];
      {
        double x1 = s.x1, x2 = s.x2;
#pragma unroll
        for (int j = 0; j < C; ++j) {
          const double xd = (double)x[j];
          ff[j] = c.b0 * xd + c.b1 * x1 + c.b2 * x2;
          x2 = x1; x1 = xd;
        }
        s.x1 = x1; s.x2 = x2;
      }
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const double yd = ff[j] - c.a1 * s.y1 - c.a2 * s.y2;
        s.y2 = s.y1; s.y1 = yd;
        float y = (float)yd;
        if (w < 1.0f) y = fmaf(y, w, x[j] * (1.0f - w));
        ptr[(size_t)(f0 + j) * n] = y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) {
        if ((uint32_t)j < c_n) {
          float y = (float)biquad_step(s, c, (double)x[j]);
          if (w < 1.0f) y = fmaf(y, w, x[j] * (1.0f - w));
          ptr[(size_t)(f0 + j) * n] = y;
        }
      }
    }
  }
  st[t] = s.x1; st[ln + t] = s.x2; st[2 * ln + t] = s.y1; st[3 * ln + t] = s.y2;
}
// The same biquad with the block cut into FOUR time segments per lane-channel (blocks of up to 256 frames).  A bank of
// thousands of lanes is a few hundred wavefronts walking 256 dependent frames each (two f64 operations of latency per
// frame, ~9 us, whatever else the chip could be doing); the recurrence is linear, so a segment can be run from a ZERO
// output state together with the two basis responses of the homogeneous part (three independent chains: they fill
// each other's latency), the four segments' true start states follow from four 2x2 products, and a second sweep runs
// each segment's recurrence again — biquad_step's operations in their order — from its true start state and stores.
// A workgroup is 64 adjacent lane-channels x 4 segments (wave = segment, so every load and store is a coalesced row);
// the hand-over goes through LDS.  The second sweep's start state agrees with the serial walk's to f64 rounding, so
// the fp32 outputs are the serial kernel's except where a value sits within 1e-16 of a rounding boundary.
constexpr int kBqSegs = 4, kBqSegMax = 64; // 4 x 64 = 256 frames
struct BqSegEnds { double z1, z2, u1, u2, v1, v2; };
// FULL: the segment holds kBqSegMax frames (straight-line code); otherwise `cnt` of them (predicated)
template <bool FULL>
__device__ __forceinline__ BqSegEnds bq_seg_sweep1(const BiquadCoefD& c, const float (&x)[kBqSegMax], uint32_t cnt, double x1, double x2) {
  BqSegEnds e{0.0, 0.0, 1.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int j = 0; j < kBqSegMax; ++j) {
    if (FULL || (uint32_t)j < cnt) {
      const double xd = (double)x[j];
      const double ff = c.b0 * xd + c.b1 * x1 + c.b2 * x2;
      const double zn = ff - c.a1 * e.z1 - c.a2 * e.z2;
      const double un = -c.a1 * e.u1 - c.a2 * e.u2, vn = -c.a1 * e.v1 - c.a2 * e.v2;
      e.z2 = e.z1; e.z1 = zn; e.u2 = e.u1; e.u1 = un; e.v2 = e.v1; e.v1 = vn;
      x2 = x1; x1 = xd;
    }
  }
  return e;
}
template <bool FULL>
__device__ __forceinline__ void bq_seg_sweep2(const BiquadCoefD& c, const float (&x)[kBqSegMax], uint32_t cnt, float w, bool live,
                                              float* __restrict__ out, uint32_t n, BiquadStateD& s) {
#pragma unroll
  for (int j = 0; j < kBqSegMax; ++j) {
    if (FULL || (uint32_t)j < cnt) {
      float y = (float)biquad_step(s, c, (double)x[j]);
      if (w < 1.0f) y = fmaf(y, w, x[j] * (1.0f - w));
      if (live) out[(size_t)j * n] = y;
    }
  }
}
__global__ __launch_bounds__(kThreads) void fx_biquad_seg_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  __shared__ double s_end[kBqSegs][6][64]; // per segment: zero-state (y1, y2) at its end, then M = [[u1, v1], [u2, v2]]
  const uint32_t lane = threadIdx.x & 63u, seg = threadIdx.x >> 6;
  const uint32_t t = blockIdx.x * 64 + lane;
  const bool live = t < 2 * n;
  const uint32_t tt = live ? t : 2 * n - 1;
  const uint32_t ch = tt / n, ln_ = tt % n;
  const size_t ln = 2 * (size_t)n;
  const uint32_t L = (frames + kBqSegs - 1) / kBqSegs;          // frames per segment (<= 64)
  const uint32_t f_lo = min(seg * L, frames), f_hi = min(f_lo + L, frames), cnt = f_hi - f_lo;
  const BiquadCoefD c{coef[ln_], coef[(size_t)n + ln_], coef[(size_t)2 * n + ln_], coef[(size_t)3 * n + ln_], coef[(size_t)4 * n + ln_]};
  const float w = wet[ln_];
  float* __restrict__ ptr = data + ch * ch_stride + ln_;
  // Everything this thread reads of the block and of the state is read BEFORE the barrier: the other segments' threads
  // overwrite the frames in front of this segment, and the last segment's thread the state, after it.
  float x[kBqSegMax];
#pragma unroll
  for (int j = 0; j < kBqSegMax; ++j) x[j] = (uint32_t)j < cnt ? ptr[(size_t)(f_lo + j) * n] : 0.0f;
  const double sx1 = st[tt], sx2 = st[ln + tt];
  double y1 = st[2 * ln + tt], y2 = st[3 * ln + tt];
  // the two inputs before the segment: the block's own frames, or the state's (x1, x2) in front of the block
  const double px1 = (f_lo >= 1 && cnt) ? (double)ptr[(size_t)(f_lo - 1) * n] : sx1;
  const double px2 = (f_lo >= 2 && cnt) ? (double)ptr[(size_t)(f_lo - 2) * n] : (f_lo == 1 ? sx1 : sx2);
  // sweep 1: zero-state response and the homogeneous basis (u: start (1, 0), v: start (0, 1)), ends only
  const BqSegEnds e = cnt == (uint32_t)kBqSegMax ? bq_seg_sweep1<true>(c, x, cnt, px1, px2) : bq_seg_sweep1<false>(c, x, cnt, px1, px2);
  s_end[seg][0][lane] = e.z1; s_end[seg][1][lane] = e.z2;
  s_end[seg][2][lane] = e.u1; s_end[seg][3][lane] = e.v1; s_end[seg][4][lane] = e.u2; s_end[seg][5][lane] = e.v2;
  __syncthreads();
  // this segment's true start state: through the earlier segments, in order
  for (uint32_t k = 0; k < seg; ++k) {
    const double n1 = s_end[k][0][lane] + s_end[k][2][lane] * y1 + s_end[k][3][lane] * y2;
    const double n2 = s_end[k][1][lane] + s_end[k][4][lane] * y1 + s_end[k][5][lane] * y2;
    y1 = n1; y2 = n2;
  }
  // sweep 2: the recurrence itself (biquad_step), from that state
  BiquadStateD s{px1, px2, y1, y2};
  if (cnt == (uint32_t)kBqSegMax) bq_seg_sweep2<true>(c, x, cnt, w, live, ptr + (size_t)f_lo * n, n, s);
  else bq_seg_sweep2<false>(c, x, cnt, w, live, ptr + (size_t)f_lo * n, n, s);
  // the segment that holds the block's last frame leaves the state
  if (live && cnt && f_hi == frames) { st[tt] = s.x1; st[ln + tt] = s.x2; st[2 * ln + tt] = s.y1; st[3 * ln + tt] = s.y2; }
}
template <int C>
__global__ __launch_bounds__(kThreads) void fx_lp24_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  const Lp24CoefD c{coef[lane], coef[(size_t)n + lane], coef[(size_t)2 * n + lane],
                    coef[(size_t)3 * n + lane], coef[(size_t)4 * n + lane], coef[(size_t)5 * n + lane]};
  Lp24StateD s{st[t], st[ln + t], st[2 * ln + t], st[3 * ln + t]};
  const float w = wet[lane];
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  // the next chunk's inputs are requested before this chunk's recurrence runs, so the memory round
  // trip of chunk k+1 hides behind the arithmetic of chunk k
  float xn[C];
#pragma unroll
  for (int j = 0; j < C; ++j) xn[j] = (uint32_t)j < frames ? ptr[(size_t)j * n] : 0.0f;
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C];
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = xn[j];
#pragma unroll
    for (int j = 0; j < C; ++j) xn[j] = f0 + C + j < frames ? ptr[(size_t)(f0 + C + j) * n] : 0.0f;
    if (c_n == (uint32_t)C) {
#pragma unroll
      for (int j = 0; j < C; ++j) {
        float y = (float)lp24_step(s, c, (double)x[j]);
        if (w < 1.0f) y = fmaf(y, w, x[j] * (1.0f - w));
        ptr[(size_t)(f0 + j) * n] = y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) {
        if ((uint32_t)j < c_n) {
          float y = (float)lp24_step(s, c, (double)x[j]);
          if (w < 1.0f) y = fmaf(y, w, x[j] * (1.0f - w));
          ptr[(size_t)(f0 + j) * n] = y;
        }
      }
    }
  }
  st[t] = s.s0; st[ln + t] = s.s1; st[2 * ln + t] = s.s2; st[3 * ln + t] = s.s3;
}

// Delay-line kinds.  Ring rows are [pos][2n] fp32 (channel-major inside a row), so the ring
// index is wave-uniform and every access is a coalesced row segment.  `w0` is the write
// index at the start of the block (host-tracked, uniform for the whole effect bank).
// a11 Delay{seconds}
template <int C>
__global__ __launch_bounds__(kThreads) void fx_delay_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    float* __restrict__ ring, uint32_t N, uint32_t w0, const float* __restrict__ wet) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  const float wm = wet[lane];
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  float* __restrict__ rg = ring + t;
  uint32_t pos = w0;
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C], y[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      uint32_t p = pos + j; if (p >= N) p -= N;
      x[j] = (uint32_t)j < c_n ? ptr[(size_t)(f0 + j) * n] : 0.0f;
      y[j] = (uint32_t)j < c_n ? rg[(size_t)p * ln] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if ((uint32_t)j < c_n) {
        uint32_t p = pos + j; if (p >= N) p -= N;
        rg[(size_t)p * ln] = x[j];
        float o = y[j];
        if (wm < 1.0f) o = fmaf(o, wm, x[j] * (1.0f - wm));
        ptr[(size_t)(f0 + j) * n] = o;
      }
    }
    pos += c_n; if (pos >= N) pos -= N;
  }
}
// a10 Chorus{voices, delay_seconds}: taps at (pos + k*spacing) mod N, k = 0..voices-1.
template <int C>
__global__ __launch_bounds__(kThreads) void fx_chorus_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    float* __restrict__ ring, uint32_t N, uint32_t w0, uint32_t voices, uint32_t spacing,
    const float* __restrict__ wet) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  const float wm = wet[lane];
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  float* __restrict__ rg = ring + t;
  uint32_t pos = w0;
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C], sum[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      x[j] = (uint32_t)j < c_n ? ptr[(size_t)(f0 + j) * n] : 0.0f;
      sum[j] = 0.0f;
    }
    for (uint32_t k = 0; k < voices; ++k) { // tap k of every frame of the chunk, loads back to back
      uint32_t base = pos + k * spacing; if (base >= N) base -= N;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        uint32_t p = base + j; if (p >= N) p -= N;
        if ((uint32_t)j < c_n) sum[j] += rg[(size_t)p * ln];
      }
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if ((uint32_t)j < c_n) {
        uint32_t p = pos + j; if (p >= N) p -= N;
        rg[(size_t)p * ln] = x[j];
        float o = sum[j];
        if (wm < 1.0f) o = fmaf(o, wm, x[j] * (1.0f - wm));
        ptr[(size_t)(f0 + j) * n] = o;
      }
    }
    pos += c_n; if (pos >= N) pos -= N;
  }
}
// a12 Reverb{attenuation, seconds}: 4 recirculating combs in parallel, 2 Schroeder all-passes
// in series.  Six rings back to back in `ring`; geometry in ReverbGeom (uniform).
struct ReverbGeom {
  uint32_t N[6];      // ring lengths (frames)
  uint32_t w[6];      // write indices at block start
  uint64_t base[6];   // row offset of each ring inside the ring buffer (rows of 2n floats)
  float g[6];         // feedback gains
};
template <int C>
__global__ __launch_bounds__(kThreads) void fx_reverb_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    float* __restrict__ ring, ReverbGeom geo, const float* __restrict__ atten, const float* __restrict__ wet) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  const float wm = wet[lane], att = atten[lane];
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  float* __restrict__ rg = ring + t;
  uint32_t pos[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pos[i] = geo.w[i];
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C], d[6][C];
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = (uint32_t)j < c_n ? ptr[(size_t)(f0 + j) * n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        uint32_t p = pos[i] + j; if (p >= geo.N[i]) p -= geo.N[i];
        d[i][j] = (uint32_t)j < c_n ? rg[(geo.base[i] + p) * ln] : 0.0f;
      }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if ((uint32_t)j < c_n) {
        const float in = x[j] * att;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t p = pos[i] + j; if (p >= geo.N[i]) p -= geo.N[i];
          const float out = geo.g[i] * d[i][j];
          rg[(geo.base[i] + p) * ln] = in + out;
          sum += out;
        }
#pragma unroll
        for (int i = 4; i < 6; ++i) {
          uint32_t p = pos[i] + j; if (p >= geo.N[i]) p -= geo.N[i];
          const float v = fmaf(geo.g[i], d[i][j], sum);
          rg[(geo.base[i] + p) * ln] = v;
          sum = fmaf(-geo.g[i], v, d[i][j]);
        }
        float o = sum;
        if (wm < 1.0f) o = fmaf(o, wm, x[j] * (1.0f - wm));
        ptr[(size_t)(f0 + j) * n] = o;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { pos[i] += c_n; if (pos[i] >= geo.N[i]) pos[i] -= geo.N[i]; }
  }
}

// ---- time-parallel forms ---------------------------------------------------------------
// A delay line that is at least one block long has no feedback INSIDE a block: every frame of
// the block reads a ring slot written in an earlier block and writes a slot nobody else in the
// block touches.  Then the effect is a pure gather/scatter over (frame, lane) and runs as one
// fully parallel, HBM-bound launch: grid = (ceil(2n / 256), frames).
// A RUN of such stages — and of the element-wise kinds, which never had feedback — is one launch: the stages of a
// chain that follow each other (config #3: chorus -> delay -> reverb combs) are applied to the element in registers,
// one after the other, exactly as the per-stage kernels would (same operations in the same order: same bits), and the
// block is read and written once instead of once per stage.  A thread owns V adjacent lane-channels of one frame
// (V = 4 when the lane count allows 16-byte accesses).  A reverb ends its run: its comb sum goes to `dst` (the
// reverb's staging block when the direct all-pass kernel follows, the block itself otherwise).
template <int V> struct VecF { float v[V]; };
template <int V> __device__ __forceinline__ VecF<V> vload(const float* p) {
  VecF<V> r;
  if constexpr (V == 4) { const float4 q = *reinterpret_cast<const float4*>(p); r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; }
  else {
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = p[j];
  }
  return r;
}
// Ring accesses of the fused run are NON-TEMPORAL: every ring row is read once and written once per trip round the ring, a ring
// length apart (hundreds of MB later), so keeping it in L2 / MALL only displaces what the chain does re-read — the block the render
// has just written, the comb sums the all-pass kernel is about to take.  Config #3, one job: 0.0507 / 0.0509 / 0.0511 -> 0.0487 /
// 0.0487 / 0.0481 ms per block (profiles/r04_ring_nt_ab.log); same bits.
typedef float groove_v4f __attribute__((ext_vector_type(4)));
template <int V> __device__ __forceinline__ VecF<V> vload_ring(const float* p) {
  VecF<V> r;
  if constexpr (V == 4) { const groove_v4f q = __builtin_nontemporal_load(reinterpret_cast<const groove_v4f*>(p)); r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; }
  else {
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = __builtin_nontemporal_load(p + j);
  }
  return r;
}
template <int V> __device__ __forceinline__ void vstore_ring(float* p, const VecF<V>& x) {
  if constexpr (V == 4) { groove_v4f q = {x.v[0], x.v[1], x.v[2], x.v[3]}; __builtin_nontemporal_store(q, reinterpret_cast<groove_v4f*>(p)); }
  else {
#pragma unroll
    for (int j = 0; j < V; ++j) __builtin_nontemporal_store(x.v[j], p + j);
  }
}
template <int V> __device__ __forceinline__ void vstore(float* p, const VecF<V>& x) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
  else {
#pragma unroll
    for (int j = 0; j < V; ++j) p[j] = x.v[j];
  }
}
constexpr int kRunMaxStages = 4;
struct FxRunStage {
  uint32_t kind, N, w, voices, spacing, pad;
  float* ring;
  const float* fa;
  const float* fb;
  const uint32_t* ua;
  const float* wet;
};
struct FxRunArgs {
  FxRunStage st[kRunMaxStages];
  ReverbGeom geo;       // of the reverb that ends the run, if one does
  const float* src;
  float* dst;
  size_t src_chs, dst_chs;
  uint32_t n_stages, n;
  float* rows;          // not null: the launch also leaves the block's lane sums, rows[wg_per_ch][2][frames] (fx_row_sum)
  uint32_t frames, wg_per_ch;
  TpPrev prev;          // groove_mix_deferred: an earlier block's lane sums, put on their bus by this launch (tp_reduce_prev)
};
// Grid of the (frame, lane-channel) effect kernels: blockIdx.y = frame, blockIdx.x = (channel, group of 256 * V lanes) —
// a workgroup never straddles the two channels, so its sum is one entry of the block's lane sums.
__host__ __device__ inline uint32_t fx_wg_per_ch(uint32_t n, uint32_t v) { return (n + kThreads * v - 1) / (kThreads * v); }
// The last launch of an effect chain hands groove_mix the block's lane sums the way the renders do (run_frames): every
// workgroup adds up what it stored and writes ONE float, rows[group][ch][frame]; the mix then reduces `wg_per_ch` short rows
// instead of reading the 8 bytes per voice-frame back (config #3: 4 rows of 2 KiB instead of 8 MiB per block).
__device__ __forceinline__ void fx_row_sum(float mine, float* __restrict__ rows, uint32_t frames, uint32_t group, uint32_t ch, uint32_t f) {
  __shared__ float red[kWaves];
  const float t = wave_sum_lane63(mine);
  if ((threadIdx.x & 63u) == 63u) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) rows[((size_t)group * 2 + ch) * frames + f] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <int V>
__device__ __forceinline__ float fx_run_element(const FxRunArgs& a, uint32_t ch, uint32_t lane, uint32_t f) {
  const uint32_t t = ch * a.n + lane;
  const size_t ln = 2 * (size_t)a.n;
  VecF<V> x = vload<V>(a.src + ch * a.src_chs + (size_t)f * a.n + lane);
#pragma unroll
  for (int s = 0; s < kRunMaxStages; ++s) {
    if ((uint32_t)s >= a.n_stages) break;
    const FxRunStage& st = a.st[s];
    VecF<V> y;
    if (st.kind == GROOVE_FX_REVERB) { // the four combs; only taken when every lane is fully wet
      const VecF<V> att = vload<V>(st.fa + lane);
      VecF<V> in;
#pragma unroll
      for (int j = 0; j < V; ++j) { in.v[j] = x.v[j] * att.v[j]; y.v[j] = 0.0f; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t p = a.geo.w[i] + f; if (p >= a.geo.N[i]) p -= a.geo.N[i];
        float* r = st.ring + (a.geo.base[i] + p) * ln + t;
        VecF<V> d = vload_ring<V>(r);
#pragma unroll
        for (int j = 0; j < V; ++j) { const float out = a.geo.g[i] * d.v[j]; d.v[j] = in.v[j] + out; y.v[j] += out; }
        vstore_ring<V>(r, d);
      }
      x = y;
      continue;
    }
    if (st.kind == GROOVE_FX_DELAY) {
      uint32_t p = st.w + f; if (p >= st.N) p -= st.N;
      float* pr = st.ring + (size_t)p * ln + t;
      y = vload_ring<V>(pr);
      vstore_ring<V>(pr, x);
    } else if (st.kind == GROOVE_FX_CHORUS) {
      uint32_t p = st.w + f; if (p >= st.N) p -= st.N;
#pragma unroll
      for (int j = 0; j < V; ++j) y.v[j] = 0.0f;
      uint32_t tp = p;
      // (the taps staged through LDS — every thread fetching its neighbour wavefront's 16 bytes, a barrier, read back — was built and
      // measured in round 3: 27.04 against 26.87 us and identical FETCH / WRITE bytes, profiles/r03_lds_staging_ab.log: every lane owns
      // its delay lines, a ring window is read exactly once by exactly one lane, there is no reuse for LDS to capture)
      for (uint32_t k = 0; k < st.voices; ++k) {
        const VecF<V> d = vload_ring<V>(st.ring + (size_t)tp * ln + t);
#pragma unroll
        for (int j = 0; j < V; ++j) y.v[j] += d.v[j];
        tp += st.spacing; if (tp >= st.N) tp -= st.N;
      }
      vstore_ring<V>(st.ring + (size_t)p * ln + t, x);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        switch (st.kind) {
          case GROOVE_FX_GAIN: y.v[j] = x.v[j] * st.fa[lane + j]; break;
          case GROOVE_FX_BITCRUSHER: y.v[j] = bitcrush(x.v[j], st.ua[lane + j]); break;
          case GROOVE_FX_LIMITER: y.v[j] = limiter(x.v[j], st.fa[lane + j], st.fb[lane + j]); break;
          case GROOVE_FX_COMPRESSOR: y.v[j] = compressor(x.v[j], st.fa[lane + j], st.fb[lane + j]); break;
          default: y.v[j] = x.v[j]; break;
        }
      }
    }
    const VecF<V> wm = vload<V>(st.wet + lane);
#pragma unroll
    for (int j = 0; j < V; ++j) x.v[j] = wm.v[j] < 1.0f ? fmaf(y.v[j], wm.v[j], x.v[j] * (1.0f - wm.v[j])) : y.v[j];
  }
  vstore<V>(a.dst + ch * a.dst_chs + (size_t)f * a.n + lane, x);
  float mine = 0.0f;
#pragma unroll
  for (int j = 0; j < V; ++j) mine += x.v[j];
  return mine;
}
template <int V>
__global__ __launch_bounds__(kThreads) void fx_run_kernel(FxRunArgs a) {
  const uint32_t ch = blockIdx.x / a.wg_per_ch, group = blockIdx.x % a.wg_per_ch;
  const uint32_t lane = (group * kThreads + threadIdx.x) * V, f = blockIdx.y;
  tp_reduce_prev(a.prev, threadIdx.x, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  // (every thread of the workgroup reaches the ONE fx_row_sum call below: it holds a barrier)
  const float mine = lane < a.n ? fx_run_element<V>(a, ch, lane, f) : 0.0f;
  if (a.rows) fx_row_sum(mine, a.rows, a.frames, group, ch, f);
}
// Reverb, stage 2: the two short Schroeder all-passes (5 ms, 1.7 ms: shorter than a block, so
// sequential per lane), chunked like the other delay-line kernels.
template <int C>
__global__ __launch_bounds__(kThreads) void fx_reverb_allpass_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride, float* __restrict__ ring, ReverbGeom geo) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, lane = t % n;
  const size_t ln = 2 * (size_t)n;
  float* __restrict__ ptr = data + ch * ch_stride + lane;
  float* __restrict__ rg = ring + t;
  uint32_t pos[2] = {geo.w[4], geo.w[5]};
  for (uint32_t f0 = 0; f0 < frames; f0 += C) {
    const uint32_t c_n = min((uint32_t)C, frames - f0);
    float x[C], d[2][C];
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = (uint32_t)j < c_n ? ptr[(size_t)(f0 + j) * n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        uint32_t p = pos[i] + j; if (p >= geo.N[4 + i]) p -= geo.N[4 + i];
        d[i][j] = (uint32_t)j < c_n ? rg[(geo.base[4 + i] + p) * ln] : 0.0f;
      }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if ((uint32_t)j < c_n) {
        float sum = x[j];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          uint32_t p = pos[i] + j; if (p >= geo.N[4 + i]) p -= geo.N[4 + i];
          const float v = fmaf(geo.g[4 + i], d[i][j], sum);
          rg[(geo.base[4 + i] + p) * ln] = v;
          sum = fmaf(-geo.g[4 + i], v, d[i][j]);
        }
        ptr[(size_t)(f0 + j) * n] = sum;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { pos[i] += c_n; if (pos[i] >= geo.N[4 + i]) pos[i] -= geo.N[4 + i]; }
  }
}

// The same two all-passes, time-parallel inside chunks: an all-pass whose line is N frames long has
// no feedback within N frames, so a chunk of N frames is a pure gather/scatter over (frame, lane) and
// only chunk k+1 depends on chunk k (through the ring).  A workgroup owns T adjacent lane-channels
// (T chosen by the host so that the launch has ~500 workgroups, and a thread issues the loads of four
// items together: the dependent chain is then 2 + 4 chunks of one memory round trip each instead of
// 256 frames / 32 per round trip), walks the
// chunks of stage 1 and then of stage 2 with a barrier between chunks; the arithmetic per frame is
// the sequential kernel's, so are the bits.  (All waves of a workgroup share one L1, so the barrier
// makes the ring and block writes of a chunk visible to the next.)
constexpr uint32_t kAllpassThreads = 1024, kAllpassUnroll = 4;
__global__ __launch_bounds__(kAllpassThreads) void fx_reverb_allpass_chunked_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride, float* __restrict__ ring, ReverbGeom geo, uint32_t T) {
  const uint32_t ln = 2 * n;
  const uint32_t t0 = blockIdx.x * T;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t N = geo.N[4 + i], w = geo.w[4 + i];
    const float g = geo.g[4 + i];
    float* __restrict__ rg = ring + geo.base[4 + i] * ln;
    for (uint32_t f0 = 0; f0 < frames; f0 += N) {
      const uint32_t items = min(N, frames - f0) * T;
      for (uint32_t k0 = threadIdx.x; k0 < items; k0 += kAllpassThreads * kAllpassUnroll) {
        float* px[kAllpassUnroll];
        float* pr[kAllpassUnroll];
        float x[kAllpassUnroll], d[kAllpassUnroll];
        bool live[kAllpassUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kAllpassUnroll; ++u) { // all the loads of this pass first: one round trip
          const uint32_t k = k0 + u * kAllpassThreads;
          const uint32_t f = f0 + k / T, tt = t0 + k % T;
          live[u] = k < items && tt < ln;
          const uint32_t ch = tt >= n ? 1u : 0u, lane = tt - ch * n;
          px[u] = data + ch * ch_stride + (size_t)f * n + lane;
          pr[u] = rg + (size_t)((w + f) % N) * ln + tt;
          x[u] = live[u] ? *px[u] : 0.0f;
          d[u] = live[u] ? *pr[u] : 0.0f;
        }
#pragma unroll
        for (uint32_t u = 0; u < kAllpassUnroll; ++u) {
          if (live[u]) {
            const float v = fmaf(g, d[u], x[u]);
            *pr[u] = v;
            *px[u] = fmaf(-g, v, d[u]);
          }
        }
      }
      __syncthreads();
    }
  }
}


// The two all-passes with no sequential step at all.  v[f] = x[f] + g v[f - N] unrolls, inside one block, into at most
// ceil(frames / N) terms that end at a ring slot written by an EARLIER block:
//     u <- ring_old[(w + f) mod N];  for j = hops-1 .. 0:  d <- u,  u <- fma(g, d, x[f - j N]);   v[f] = u,  out[f] = fma(-g, u, d)
// which is the sequential kernel's chain of operations for that frame, evaluated from its oldest term: same bits.  The
// second all-pass takes the first one's output at f, f - N2, f - 2 N2 ... and evaluates each of those the same way
// (<= 4 x 2 input reads for 1.7 ms / 5 ms lines and 256 frames, almost all L2 hits: neighbouring frames read the same
// rows).  Every (frame, lane-channel) is independent, so the launch is (lane-channels / V) x frames threads and one
// memory round trip deep, against 2 + 4 dependent chunk passes.  Reads and writes must not alias: the input comes from
// a staging block `src` (the comb sum), the result goes to the block, and the rings are double-buffered (`old_base`
// read, `new_base` written; rows of a ring that this block does not rewrite are copied over by the rows of the grid
// beyond `frames`).  The host swaps the bases afterwards.
struct AllpassDirectArgs {
  const float* src;
  float* dst;
  float* ring;
  size_t src_chs, dst_chs;
  uint64_t old_base[2], new_base[2];
  uint32_t N[2], w[2];
  float g[2];
  uint32_t n, frames;
  float* rows;          // not null: the block's lane sums, rows[wg_per_ch][2][frames] (fx_row_sum)
  uint32_t wg_per_ch;
  TpPrev prev;          // the all-pass stream (groove_set_fx_allpass_stream): the previous block's lane sums, put on their bus by this launch
};
template <int V>
__device__ __forceinline__ float allpass_direct_element(const AllpassDirectArgs& a, uint32_t ch, uint32_t lane, uint32_t f) {
  const uint32_t t = ch * a.n + lane;
  const size_t ln = 2 * (size_t)a.n;
  if (f >= a.frames) { // ring rows this block leaves as they are
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (f < a.N[i]) {
        const uint32_t slot = (a.w[i] + f) % a.N[i];
        vstore<V>(a.ring + (a.new_base[i] + slot) * ln + t, vload<V>(a.ring + (a.old_base[i] + slot) * ln + t));
      }
    return 0.0f;
  }
  const float* __restrict__ x = a.src + ch * a.src_chs + lane;
  const uint32_t slot1 = (a.w[1] + f) % a.N[1];
  VecF<V> u = vload<V>(a.ring + (a.old_base[1] + slot1) * ln + t), d = u, v0;
  for (int j = (int)(f / a.N[1]); j >= 0; --j) {
    const uint32_t np = f - (uint32_t)j * a.N[1]; // the first all-pass, at frame np
    VecF<V> ua = vload<V>(a.ring + (a.old_base[0] + (a.w[0] + np) % a.N[0]) * ln + t), da = ua;
    for (int i = (int)(np / a.N[0]); i >= 0; --i) {
      const VecF<V> xin = vload<V>(x + (size_t)(np - (uint32_t)i * a.N[0]) * a.n);
      da = ua;
#pragma unroll
      for (int k = 0; k < V; ++k) ua.v[k] = fmaf(a.g[0], da.v[k], xin.v[k]);
    }
    if (j == 0) v0 = ua;
    d = u;
#pragma unroll
    for (int k = 0; k < V; ++k) u.v[k] = fmaf(a.g[1], d.v[k], fmaf(-a.g[0], ua.v[k], da.v[k]));
  }
  VecF<V> out;
#pragma unroll
  for (int k = 0; k < V; ++k) out.v[k] = fmaf(-a.g[1], u.v[k], d.v[k]);
  vstore<V>(a.dst + ch * a.dst_chs + (size_t)f * a.n + lane, out);
  if (f + a.N[0] >= a.frames) vstore<V>(a.ring + (a.new_base[0] + (a.w[0] + f) % a.N[0]) * ln + t, v0);
  if (f + a.N[1] >= a.frames) vstore<V>(a.ring + (a.new_base[1] + slot1) * ln + t, u);
  float mine = 0.0f;
#pragma unroll
  for (int k = 0; k < V; ++k) mine += out.v[k];
  return mine;
}
template <int V>
__global__ __launch_bounds__(kThreads) void fx_reverb_allpass_direct_kernel(AllpassDirectArgs a) {
  tp_reduce_prev(a.prev, threadIdx.x, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const uint32_t ch = blockIdx.x / a.wg_per_ch, group = blockIdx.x % a.wg_per_ch;
  const uint32_t lane = (group * kThreads + threadIdx.x) * V, f = blockIdx.y;
  const float mine = lane < a.n ? allpass_direct_element<V>(a, ch, lane, f) : 0.0f;
  if (a.rows && f < a.frames) fx_row_sum(mine, a.rows, a.frames, group, ch, f); // (uniform in the workgroup: it holds a barrier)
}

#endif // GROOVE_WELSH_CLASS_TU

} // namespace groove
