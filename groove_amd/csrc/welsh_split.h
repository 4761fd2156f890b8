// welsh_split.h — the ROLE-SPLIT form of the wave-uniform Welsh render (mid-size banks).
//
// One voice per lane leaves a bank that does not fill the chip waiting for ONE wavefront's serial walk of the block: a lone
// wavefront issues one instruction per ~9 cycles WHATEVER the instruction mix or its instruction-level parallelism
// (docs/VALU_COSTS.md: four independent chains in one wave issue no faster), so a 256-frame block of a retuning patch
// (~90 instructions per frame) takes ~0.11 ms on a SIMD that could issue four times as much.  More voices per SIMD would use
// the slack — a 65,536-voice bank has none to offer.  The time-parallel form (welsh_tp.h) has no such floor but costs ~4x
// the work per voice and loses above ~24,000 voices.
//
// What is left is to give ONE voice-wave's frame to SEVERAL wavefronts.  A frame of a Welsh voice is a feed-forward front
// (envelopes, LFO, oscillators, cutoff percent), the filter coefficients' tangent, and the f64 filter recurrence with the
// output gains; nothing flows backwards (dsp_core.h: welsh_frame = FRONT / COEF / BACK).  Three wavefronts per 64 voices
// form a pipeline over the block's frames, kSplitChunk frames per step, through LDS:
//     A  front:  frames of chunk c     -> {sum, gain} and the cutoff percent
//     B  mid:    chunk c - 1: tangent of the cutoff (exp2, polynomial); every eighth frame it also turns the bus tile
//     C  back:   chunk c - 2: coefficients from the tangent, filter step, gains -> bus tile (and the planar block)
// one workgroup barrier per step.  Every quantity is computed by the statements of the serial kernels in their order, so
// the results — bus rows, blocks, state — are the serial kernels' BIT FOR BIT (tests/test_gpu_split.py).  The walk of a
// block is then as long as its longest role (~40 of ~90 instructions per frame) instead of their sum.
//
// Scope: the four class-specialised base kinds (f32 / smooth-f64 LFO x static / retuned filter); workgroups of the two
// exact-f64 kinds (rare) keep the all-kinds kernel.  Workgroup = 4 virtual waves x 3 roles = 768 threads, 72 KB of LDS,
// one workgroup per CU (128 VGPRs): a role of each kind on every SIMD.
#pragma once
#include "kernels.h"

namespace groove {

constexpr int kSplitVw = kWaves;                       // virtual waves per workgroup (the host's workgroup = 4 virtual waves)
constexpr int kSplitLanes = kSplitVw * 64;             // 256 voices
constexpr int kSplitThreads = 3 * kSplitLanes;         // roles A, B, C
constexpr uint32_t kSplitChunk = 4;                    // frames per pipeline step (measured: 8 — half the barriers — is 30 % slower)
constexpr uint32_t kSplitGroup = 8;                    // frames per turn of the bus tile (FusedAcc::kChunk)
static_assert(kSplitGroup % kSplitChunk == 0 && kSplitGroup == FusedAcc::kChunk, "the bus tile is turned every second step");

struct SplitLds {
  float2 ac[3][kSplitChunk][kSplitLanes];  // A -> C: {sum (NaN: the lane is silent this frame), gain}; three steps deep
  float ab[2][kSplitChunk][kSplitLanes];   // A -> B: cutoff percent (NaN: no retune this frame)
  float bc[2][kSplitChunk][kSplitLanes];   // B -> C: tan of the cutoff, negated above SR/4 (NaN: coefficients stand)
  float2 tile[2][kSplitGroup][kSplitLanes]; // C -> B: (L, R) of eight frames, two groups in rotation
};
static_assert(sizeof(SplitLds) <= 72 * 1024, "fits beside another workgroup's");

// Word ranges of WelshState (dsp_core.h): [0, 30) oscillators, increments, envelopes — role A; [30, 38) the filter — role C;
// [38, 40) flags — role A.
constexpr uint32_t kStateFiltWord = offsetof(WelshState, filt) / 4, kStateFlagsWord = offsetof(WelshState, vflags) / 4;
static_assert(kStateFiltWord == 30 && kStateFlagsWord == 38 && sizeof(WelshState) == 160, "role A / role C state words");
template <class T>
__device__ __forceinline__ void soa_store_range(uint32_t* __restrict__ buf, uint32_t n, uint32_t v, const T& x, uint32_t w0, uint32_t w1) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)(sizeof(T) / 4 * n * 4u), 0x00020000);
  const WordsOf<T> t = __builtin_bit_cast(WordsOf<T>, x);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4; ++i)
    if (i >= w0 && i < w1) __builtin_amdgcn_raw_buffer_store_b32((int)t.w[i], rsrc, (int)(v * 4u), (int)(i * n * 4u), 0);
}

// What every role needs of its virtual wave.
struct SplitWave { WaveDesc d; uint32_t wg, l, v; bool active; };
__device__ __forceinline__ SplitWave split_wave(UniformArgsPtr a, uint32_t local /* 0 .. 255 within the role */) {
  SplitWave w;
  w.wg = a->wg_list[GROOVE_WG_SLOT(a->n_wgs)];
  w.l = local;
  const uint32_t lane = local & 63u;
  const uint32_t w0 = w.wg * kSplitVw + (local >> 6);
  const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(w0, a->n_waves - 1));
  w.d = make_scalar(a->waves[wi]);
  w.active = (w0 < a->n_waves) && (lane < w.d.count);
  w.v = w.active ? w.d.vbase + lane : w.d.vbase; // idle lanes shadow the run's first voice
  return w;
}
// Length of a new boundary-free segment at frame f: the wave minimum over the ACTIVE lanes (shadow lanes may hold a torn
// record — kernels.h run_frames_segmented), at least one frame, counted if the minimum is 0 all the same (diag.h).
__device__ __forceinline__ uint32_t split_segment(UniformArgsPtr a, const SplitWave& w, const WelshState& s, uint32_t mine, uint32_t f, uint32_t frames) {
#ifdef GROOVE_DIAG_SHADOW_IN_MIN
  const uint32_t wmin = wave_min_u32(mine);
#else
  const uint32_t wmin = wave_min_u32(w.active ? mine : 0xFFFFFFFFu);
#endif
  if (wmin == 0) welsh_diag_zero(DiagWhere{a->diag, w.wg, w.wg * kSplitVw + (w.l >> 6), w.d.count}, s, w.active, f, mine);
  return max(1u, min(wmin, frames - f));
}
__device__ __forceinline__ SplitLds& split_lds() {
  __shared__ SplitLds lds;
  return lds;
}
// ROLES = 3: front | tangent (+ tile turn) | back, as above.  ROLES = 2: front + tangent | back (+ tile turn): eight wavefronts per
// workgroup instead of twelve, so TWO workgroups fit a CU (4 waves per SIMD, 128 VGPRs) and banks of up to 131,072 voices run
// in one round; the front role is then the longer one (it carries the tangent too).
template <int ROLES> __device__ __forceinline__ uint32_t split_steps(uint32_t frames) { return (frames + kSplitChunk - 1) / kSplitChunk + (ROLES - 1); }

#ifdef GROOVE_SPLIT_PROBE /* measurement build only (tools/split_probe.sh): cycles every role spends on its step and at the barrier */
static __device__ unsigned long long g_split_probe[4][3]; // [role][busy, wait, waves]
#define SPLIT_PROBE_BEGIN uint64_t pb_busy = 0, pb_wait = 0, pb_t0 = 0, pb_t1 = 0;
#define SPLIT_PROBE_STEP pb_t0 = __builtin_amdgcn_s_memtime();
#define SPLIT_PROBE_BARRIER pb_t1 = __builtin_amdgcn_s_memtime(); pb_busy += pb_t1 - pb_t0;
#define SPLIT_PROBE_AFTER pb_wait += __builtin_amdgcn_s_memtime() - pb_t1;
#define SPLIT_PROBE_END(role) if ((threadIdx.x & 63u) == 0) { atomicAdd(&g_split_probe[role][0], pb_busy); atomicAdd(&g_split_probe[role][1], pb_wait); atomicAdd(&g_split_probe[role][2], 1ull); }
#else
#define SPLIT_PROBE_BEGIN
#define SPLIT_PROBE_STEP
#define SPLIT_PROBE_BARRIER
#define SPLIT_PROBE_AFTER
#define SPLIT_PROBE_END(role)
#endif

// ---- role A: the front of every frame of the block (run_frames_segmented's walk, one chunk per step)
template <int ROLES, int LFO_MODE, bool RETUNE, int C1, int C2, int CL>
__device__ __forceinline__ void welsh_split_front_impl(UniformArgsPtr a) {
  constexpr bool REST = LFO_MODE != LFO_F64;
  constexpr bool WITH_TAN = ROLES == 2 && RETUNE; // two roles: this one also takes the tangent of the cutoff
  SplitLds& lds = split_lds();
  const SplitWave w = split_wave(a, threadIdx.x);
  const uint32_t n = a->n, frames = a->frames;
  const WelshParams& p = w.d.p;
  WelshState s = soa_load<WelshState>(a->state, n, w.v);
  WelshScratch sc;
  sc.prev_pct = 0.0f; sc.ls = 0.0; sc.lc = 1.0; sc.lm = 1.0; sc.ta = 0.0f; sc.tf = 0.0f; // (coefficients: role C's business)
  RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi};
  if constexpr (WITH_TAN) asm volatile("" : "+v"(rc.tan_k1), "+v"(rc.tan_k2), "+v"(rc.log2_x0), "+v"(rc.x_hi));
  const uint32_t steps = split_steps<ROLES>(frames), nch = steps - (ROLES - 1);
  uint32_t seg_left = 0, seg_len = 0;
  bool live = false;
  const float kNan = __builtin_nanf("");
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (it < nch) {
      const uint32_t f0 = it * kSplitChunk, cnt = min(kSplitChunk, frames - f0);
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) if (j < cnt) {
        const uint32_t f = f0 + j;
        {
          float sum = 0.0f, g = 0.0f, pct = 0.0f, lfo = 0.0f;
          bool retune = false, ok;
          if (f == 0) { // the checked form (first tick after a note event, envelope boundaries, the idle test)
            ok = welsh_frame_front<true, RETUNE, LFO_MODE, C1, C2, CL, false, REST>(p, s, sc, sum, g, pct, retune, lfo) && w.active;
          } else {
            if (seg_left == 0) { // a new boundary-free segment (kernels.h run_frames_segmented)
              const uint32_t mine = welsh_segment_begin(p, s, live);
              welsh_segment_start_hoisted(s, sc);
              live = live && w.active;
              seg_len = seg_left = split_segment(a, w, s, mine, f, frames);
            }
            ok = live;
            if (live) welsh_frame_front<false, RETUNE, LFO_MODE, C1, C2, CL, true, REST, true>(p, s, sc, sum, g, pct, retune, lfo);
            if (--seg_left == 0) welsh_segment_end_hoisted<CL == LFO_UNUSED>(p, s, seg_len, live);
          }
          lds.ac[it % 3][j][w.l] = make_float2(ok ? sum : kNan, g);
          if constexpr (WITH_TAN) { // role B's statements (welsh_split_mid), in line
            float t = kNan;
            if (ok && retune) { bool hi; const float tj = lp24_t_from_pct(pct, rc, hi); t = hi ? -tj : tj; }
            lds.bc[it & 1][j][w.l] = t;
          } else if (RETUNE) lds.ab[it & 1][j][w.l] = (ok && retune) ? pct : kNan;
        }
      }
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(0)
  if (w.active) {
    soa_store_range(a->state, n, w.v, s, 0, kStateFiltWord);
    soa_store_range(a->state, n, w.v, s, kStateFlagsWord, (uint32_t)(sizeof(WelshState) / 4));
  }
}
template <int ROLES, int LFO_MODE, bool RETUNE, int C1, int C2, int CL>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split_front(UniformArgsPtr a) {
  welsh_split_front_impl<ROLES, LFO_MODE, RETUNE, C1, C2, CL>(uniform_args_scalar(a));
}

// ---- role B: the tangent of the cutoff, one step behind A; and the bus tile's turn (FusedAcc::flush, on the
// group of eight frames role C finished in the previous step)
template <class Lds>
__device__ __forceinline__ void split_turn_tile(const Lds& lds, uint32_t l, uint32_t f_end, float* __restrict__ rows, uint32_t wg, uint32_t frames) {
  const uint32_t f_lo = (f_end - 1) / kSplitGroup * kSplitGroup, count = f_end - f_lo, g = (f_lo / kSplitGroup) & 1u;
  const uint32_t row = l >> 5, col = l & 31u; // 32 lanes per frame row, as FusedAcc::flush
  const float2* __restrict__ src = &lds.tile[g][0][0] + row * kSplitLanes + col;
  float sl = 0.0f, sr = 0.0f;
#pragma unroll
  for (uint32_t j = 0; j < kSplitLanes / 32; ++j) { const float2 v = src[j * 32]; sl += v.x; sr += v.y; }
  sl = dpp_add<0xb1, 0xf>(sl); sr = dpp_add<0xb1, 0xf>(sr);
  sl = dpp_add<0x4e, 0xf>(sl); sr = dpp_add<0x4e, 0xf>(sr);
  sl = dpp_add<0x124, 0xf>(sl); sr = dpp_add<0x124, 0xf>(sr);
  sl = dpp_add<0x128, 0xf>(sl); sr = dpp_add<0x128, 0xf>(sr);
  sl = dpp_add<0x142, 0xa>(sl); sr = dpp_add<0x142, 0xa>(sr);
  if (col == 31 && row < count) { // lanes 31 and 63 of each wave hold the totals of their rows
    rows[((size_t)wg * 2 + 0) * frames + f_lo + row] = sl;
    rows[((size_t)wg * 2 + 1) * frames + f_lo + row] = sr;
  }
}
template <bool RETUNE>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split_mid(UniformArgsPtr ka) {
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds& lds = split_lds();
  const uint32_t l = threadIdx.x - kSplitLanes;
  const uint32_t frames = a->frames, wg = a->wg_list[GROOVE_WG_SLOT(a->n_wgs)];
  RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi};
  if constexpr (RETUNE) asm volatile("" : "+v"(rc.tan_k1), "+v"(rc.tan_k2), "+v"(rc.log2_x0), "+v"(rc.x_hi));
  float* __restrict__ rows = a->rows;
  const uint32_t steps = split_steps<3>(frames), nch = steps - 2;
  const float kNan = __builtin_nanf("");
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (RETUNE && it >= 1 && it <= nch) {
      const uint32_t c = it - 1;
      // the step's eight percents first (one LDS round trip, not eight), then eight independent tangents
      float pct[kSplitChunk], t[kSplitChunk];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) pct[j] = lds.ab[c & 1][j][l]; // (frames past the block: stale values, harmless)
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        bool hi;
        const float tj = lp24_t_from_pct(pct[j], rc, hi); // (branch-free: a NaN percent — no retune this frame — is sorted out by the select)
        t[j] = (pct[j] == pct[j]) ? (hi ? -tj : tj) : kNan; // t > 0 always: the sign carries the side of SR/4
      }
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) lds.bc[c & 1][j][l] = t[j];
    }
    // role C wrote chunk it - 3 in the PREVIOUS step: when that chunk completed a group of eight frames, the group is turned
    // now (role C is writing the other tile buffer meanwhile)
    if (it >= 3) {
      const uint32_t f_end = (it - 2) * kSplitChunk; // end of chunk it - 3; a chunk inside the loop is never the block's last
      if ((f_end % kSplitGroup) == 0) split_turn_tile(lds, l, f_end, rows, wg, frames);
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(1)
  split_turn_tile(lds, l, frames, rows, wg, frames); // the group that holds the block's last frame (role C's last step)
}

// ---- role C: coefficients from the tangent, the filter recurrence, the gains; two steps behind A
template <int ROLES, bool FUSED, bool RETUNE>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split_back(UniformArgsPtr ka) {
  constexpr uint32_t LAG = ROLES - 1; // steps behind the front role
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds& lds = split_lds();
  const SplitWave w = split_wave(a, threadIdx.x - (ROLES - 1) * kSplitLanes);
  const uint32_t n = a->n, frames = a->frames;
  const WelshParams& p = w.d.p;
  const RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi};
  Lp24StateD filt; // the filter's words of the state record, nothing else
  {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a->state, 0, (int)(sizeof(WelshState) / 4 * n * 4u), 0x00020000);
    WordsOf<Lp24StateD> fw;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Lp24StateD) / 4; ++i)
      fw.w[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(w.v * 4u), (int)((kStateFiltWord + i) * n * 4u), 0);
    filt = __builtin_bit_cast(Lp24StateD, fw);
  }
  Lp24CoefD coef = lp24_coefd_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max); // welsh_scratch_init
  if (!RETUNE) coef = make_scalar(coef);
  float* __restrict__ out = a->out;
  const size_t chs = a->ch_stride;
  const uint32_t steps = split_steps<ROLES>(frames);
  float* __restrict__ rows = a->rows;
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (ROLES == 2 && it >= 2) { // two roles: this one turns the bus tile — the group its PREVIOUS step completed, if it did
      const uint32_t f_end = (it - 1) * kSplitChunk; // end of chunk it - 2
      if ((f_end % kSplitGroup) == 0) split_turn_tile(lds, w.l, f_end, rows, w.wg, frames);
    }
    if (it >= LAG) {
      const uint32_t c = it - LAG, f0 = c * kSplitChunk, cnt = min(kSplitChunk, frames - f0);
      // the step's inputs first: one LDS round trip for the eight frames instead of two per frame
      float2 in[kSplitChunk];
      float tt[kSplitChunk];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        in[j] = lds.ac[c % 3][j][w.l];
        tt[j] = RETUNE ? lds.bc[c & 1][j][w.l] : 0.0f;
      }
      float2* __restrict__ tile = &lds.tile[(f0 / kSplitGroup) & 1u][f0 & (kSplitGroup - 1)][w.l];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        if (j < cnt) {
          const uint32_t f = f0 + j;
          float L = 0.0f, R = 0.0f;
          if (in[j].x == in[j].x) { // the voice sounds on this frame
            if (RETUNE) {
              if (tt[j] == tt[j]) coef = lp24_coefd_from_t(p.fc, fabsf(tt[j]), tt[j] < 0.0f, (p.flags & WF_COEF_WIDE) != 0);
            }
            welsh_frame_back<!RETUNE>(p, filt, coef, in[j].x, in[j].y, L, R);
          }
          tile[j * kSplitLanes] = make_float2(L, R);
          if (!FUSED && w.active) {
            block_store(out + (size_t)f * n + w.v, L);
            block_store(out + chs + (size_t)f * n + w.v, R);
          }
        }
      }
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(2)
  if (ROLES == 2) split_turn_tile(lds, w.l, frames, rows, w.wg, frames); // the group that holds the block's last frame
  if (w.active) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a->state, 0, (int)(sizeof(WelshState) / 4 * n * 4u), 0x00020000);
    const WordsOf<Lp24StateD> fw = __builtin_bit_cast(WordsOf<Lp24StateD>, filt);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Lp24StateD) / 4; ++i)
      __builtin_amdgcn_raw_buffer_store_b32((int)fw.w[i], rsrc, (int)(w.v * 4u), (int)((kStateFiltWord + i) * n * 4u), 0);
  }
}

// ---- FOUR roles (round 3): the front in two halves and the coefficients' fp32 quotients in the tangent's role.
// Measured on the three-role form (tools/split_probe.py, cycles per role between barriers): the tangent's role is busy a third
// of a step, while the front (pitch / pulse-width LFO kinds: the f64 LFO, two u64 <-> f64 round trips, the oscillators) or the
// back (retuned kinds: two reciprocals and four widenings before the ten f64 operations of the filter) set the step's length.
//     A1 ctl:  chunk c      envelopes, LFO -> gain (to C), cutoff percent (to B), `mod` (to A2; NaN: the lane is silent)
//     A2 osc:  chunk c - 1  the oscillators under `mod` -> sum (to C; NaN: silent)
//     B  mid:  chunk c - 1  tangent of the cutoff AND the coefficients' fp32 quotients (to C); the bus tile's turn
//     C  back: chunk c - 2  widens the quotients, filter step, gains -> bus tile (and the planar block)
// Sixteen wavefronts per workgroup (1,024 threads, 124 KB of LDS, one workgroup per CU): four per SIMD, one of each role.
// welsh_frame_ctl / welsh_frame_osc / lp24_coefq_from_t / lp24_coefd_from_q (dsp_core.h) are the serial statements cut at
// those points, so the results stay the serial kernels' bit for bit.
struct SplitLds4 {
  double mod[2][kSplitChunk][kSplitLanes];   // A1 -> A2
  float gain[3][kSplitChunk][kSplitLanes];   // A1 -> C, three steps deep
  float pct[2][kSplitChunk][kSplitLanes];    // A1 -> B (NaN: no retune this frame)
  float sum[2][kSplitChunk][kSplitLanes];    // A2 -> C
  float4 q4[2][kSplitChunk][kSplitLanes];    // B -> C: {b0 (NaN: coefficients stand), q2 (negated above SR/4), b0', q2'} of the two sections
  float2 q2[2][kSplitChunk][kSplitLanes];    // B -> C: {P / D', P' / D''} (upper side)
  float2 tile[2][kSplitGroup][kSplitLanes];  // C -> B
};
static_assert(sizeof(SplitLds4) <= 128 * 1024, "one workgroup per CU");
__device__ __forceinline__ SplitLds4& split_lds4() {
  __shared__ SplitLds4 lds;
  return lds;
}
constexpr uint32_t kStateLfoWord = offsetof(WelshState, lfo) / 4, kStateIncWord = offsetof(WelshState, o1_inc) / 4, kStateEnvWord = offsetof(WelshState, amp) / 4;
static_assert(kStateLfoWord == 8 && kStateIncWord == 12 && kStateEnvWord == 16, "role A1 / role A2 state words");

// role A1 (run_frames_segmented's walk, as welsh_split_front_impl)
template <int LFO_MODE, bool RETUNE, int CL>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split4_ctl(UniformArgsPtr ka) {
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds4& lds = split_lds4();
  const SplitWave w = split_wave(a, threadIdx.x);
  const uint32_t n = a->n, frames = a->frames;
  const WelshParams& p = w.d.p;
  WelshState s = soa_load<WelshState>(a->state, n, w.v);
  WelshScratch sc;
  sc.prev_pct = 0.0f; sc.ls = 0.0; sc.lc = 1.0; sc.lm = 1.0; sc.ta = 0.0f; sc.tf = 0.0f;
  const uint32_t steps = split_steps<3>(frames), nch = steps - 2;
  uint32_t seg_left = 0, seg_len = 0;
  bool live = false;
  const float kNan = __builtin_nanf("");
  const double kNanD = __builtin_nan("");
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (it < nch) {
      const uint32_t f0 = it * kSplitChunk, cnt = min(kSplitChunk, frames - f0);
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) if (j < cnt) {
        const uint32_t f = f0 + j;
        float g = 0.0f, pct = 0.0f;
        double mod = 0.0;
        bool retune = false, first = false, ok;
        if (f == 0) {
          ok = welsh_frame_ctl<true, RETUNE, LFO_MODE, CL, false, false>(p, s, sc, g, pct, retune, mod, first) && w.active;
        } else {
          if (seg_left == 0) {
            const uint32_t mine = welsh_segment_begin(p, s, live);
            welsh_segment_start_hoisted(s, sc);
            live = live && w.active;
            seg_len = seg_left = split_segment(a, w, s, mine, f, frames);
          }
          ok = live;
          if (live) welsh_frame_ctl<false, RETUNE, LFO_MODE, CL, true, true>(p, s, sc, g, pct, retune, mod, first);
          if (--seg_left == 0) welsh_segment_end_hoisted<CL == LFO_UNUSED>(p, s, seg_len, live);
        }
        lds.mod[it & 1][j][w.l] = ok ? mod : kNanD;
        lds.gain[it % 3][j][w.l] = g;
        if (RETUNE) lds.pct[it & 1][j][w.l] = (ok && retune) ? pct : kNan;
      }
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(0)
  if (w.active) {
    soa_store_range(a->state, n, w.v, s, kStateLfoWord, kStateIncWord);
    soa_store_range(a->state, n, w.v, s, kStateEnvWord, kStateFiltWord);
    soa_store_range(a->state, n, w.v, s, kStateFlagsWord, (uint32_t)(sizeof(WelshState) / 4));
  }
}
// role A2: one step behind A1
template <int LFO_MODE, int C1, int C2>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split4_osc(UniformArgsPtr ka) {
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds4& lds = split_lds4();
  const SplitWave w = split_wave(a, threadIdx.x - kSplitLanes);
  const uint32_t n = a->n, frames = a->frames;
  const WelshParams& p = w.d.p;
  WelshState s = soa_load<WelshState>(a->state, n, w.v); // (the oscillators' words, the base increments and the flags are all it uses)
  const bool first0 = (s.vflags & VF_FIRST) != 0;
  const uint32_t steps = split_steps<3>(frames), nch = steps - 2;
  const float kNan = __builtin_nanf("");
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (it >= 1 && it <= nch) {
      const uint32_t c = it - 1, f0 = c * kSplitChunk, cnt = min(kSplitChunk, frames - f0);
      double mod[kSplitChunk];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) mod[j] = lds.mod[c & 1][j][w.l];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) if (j < cnt) {
        float sum = kNan;
        if (mod[j] == mod[j]) sum = welsh_frame_osc<LFO_MODE, C1, C2, true>(p, s, true, mod[j], f0 + j == 0 && first0);
        lds.sum[c & 1][j][w.l] = sum;
      }
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(1)
  if (w.active) soa_store_range(a->state, n, w.v, s, 0, kStateLfoWord);
}
// role B
template <bool RETUNE>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split4_mid(UniformArgsPtr ka) {
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds4& lds = split_lds4();
  const SplitWave w = split_wave(a, threadIdx.x - 2 * kSplitLanes);
  const uint32_t l = w.l;
  const WelshParams& p = w.d.p;
  const uint32_t frames = a->frames, wg = w.wg;
  RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi};
  if constexpr (RETUNE) asm volatile("" : "+v"(rc.tan_k1), "+v"(rc.tan_k2), "+v"(rc.log2_x0), "+v"(rc.x_hi));
  float* __restrict__ rows = a->rows;
  const uint32_t steps = split_steps<3>(frames), nch = steps - 2;
  const float kNan = __builtin_nanf("");
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (RETUNE && it >= 1 && it <= nch) {
      const uint32_t c = it - 1;
      float pct[kSplitChunk];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) pct[j] = lds.pct[c & 1][j][l];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        float4 q4 = make_float4(kNan, 0.0f, 0.0f, 0.0f);
        float2 q2 = make_float2(0.0f, 0.0f);
        if (pct[j] == pct[j]) {
          bool hi;
          const float t = lp24_t_from_pct(pct[j], rc, hi);
          const Lp24CoefQ q = lp24_coefq_from_t(p.fc, t, hi, (p.flags & WF_COEF_WIDE) != 0);
          q4 = make_float4(q.ba, hi ? -q.qa : q.qa, q.bb, q.qb); // q2 > 0 always: the sign carries the side of SR/4
          q2 = make_float2(q.pa, q.pb);
        }
        lds.q4[c & 1][j][l] = q4;
        lds.q2[c & 1][j][l] = q2;
      }
    }
    if (it >= 3) {
      const uint32_t f_end = (it - 2) * kSplitChunk;
      if ((f_end % kSplitGroup) == 0) split_turn_tile(lds, l, f_end, rows, wg, frames);
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(2)
  split_turn_tile(lds, l, frames, rows, wg, frames);
}
// role C
template <bool FUSED, bool RETUNE>
GROOVE_BODY_LINKAGE __device__ __attribute__((noinline)) void welsh_split4_back(UniformArgsPtr ka) {
  const UniformArgsPtr a = uniform_args_scalar(ka);
  SplitLds4& lds = split_lds4();
  const SplitWave w = split_wave(a, threadIdx.x - 3 * kSplitLanes);
  const uint32_t n = a->n, frames = a->frames;
  const WelshParams& p = w.d.p;
  const RenderConsts rc{a->rc.pi_over_sr, a->rc.fc_max, a->rc.log2_x0, a->rc.x_lo, a->rc.x_hi};
  Lp24StateD filt;
  {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a->state, 0, (int)(sizeof(WelshState) / 4 * n * 4u), 0x00020000);
    WordsOf<Lp24StateD> fw;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Lp24StateD) / 4; ++i)
      fw.w[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(w.v * 4u), (int)((kStateFiltWord + i) * n * 4u), 0);
    filt = __builtin_bit_cast(Lp24StateD, fw);
  }
  Lp24CoefD coef = lp24_coefd_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max);
  if (!RETUNE) coef = make_scalar(coef);
  float* __restrict__ out = a->out;
  const size_t chs = a->ch_stride;
  const uint32_t steps = split_steps<3>(frames);
  SPLIT_PROBE_BEGIN
  for (uint32_t it = 0; it < steps; ++it) {
    SPLIT_PROBE_STEP
    if (it >= 2) {
      const uint32_t c = it - 2, f0 = c * kSplitChunk, cnt = min(kSplitChunk, frames - f0);
      float sum[kSplitChunk], gain[kSplitChunk];
      float4 q4[kSplitChunk];
      float2 q2[kSplitChunk];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        sum[j] = lds.sum[c & 1][j][w.l];
        gain[j] = lds.gain[c % 3][j][w.l];
        if (RETUNE) { q4[j] = lds.q4[c & 1][j][w.l]; q2[j] = lds.q2[c & 1][j][w.l]; }
      }
      float2* __restrict__ tile = &lds.tile[(f0 / kSplitGroup) & 1u][f0 & (kSplitGroup - 1)][w.l];
#pragma unroll
      for (uint32_t j = 0; j < kSplitChunk; ++j) {
        if (j < cnt) {
          const uint32_t f = f0 + j;
          float L = 0.0f, R = 0.0f;
          if (sum[j] == sum[j]) {
            if (RETUNE) {
              if (q4[j].x == q4[j].x) coef = lp24_coefd_from_q(Lp24CoefQ{q4[j].x, fabsf(q4[j].y), q4[j].z, q4[j].w, q2[j].x, q2[j].y}, q4[j].y < 0.0f, (p.flags & WF_COEF_WIDE) != 0);
            }
            welsh_frame_back<!RETUNE>(p, filt, coef, sum[j], gain[j], L, R);
          }
          tile[j * kSplitLanes] = make_float2(L, R);
          if (!FUSED && w.active) {
            block_store(out + (size_t)f * n + w.v, L);
            block_store(out + chs + (size_t)f * n + w.v, R);
          }
        }
      }
    }
    SPLIT_PROBE_BARRIER
    __syncthreads();
    SPLIT_PROBE_AFTER
  }
  SPLIT_PROBE_END(3)
  if (w.active) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a->state, 0, (int)(sizeof(WelshState) / 4 * n * 4u), 0x00020000);
    const WordsOf<Lp24StateD> fw = __builtin_bit_cast(WordsOf<Lp24StateD>, filt);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Lp24StateD) / 4; ++i)
      __builtin_amdgcn_raw_buffer_store_b32((int)fw.w[i], rsrc, (int)(w.v * 4u), (int)((kStateFiltWord + i) * n * 4u), 0);
  }
}

// A workgroup whose voices are all silent with both envelopes idle writes its zero rows and leaves (kernels.h
// welsh_idle_workgroup); every role-wave looks at its own virtual wave.
__device__ __forceinline__ bool welsh_split_idle_workgroup(const UniformArgs& a, uint32_t threads) {
  const uint32_t wg = a.wg_list[GROOVE_WG_SLOT(a.n_wgs)];
  const uint32_t local = threadIdx.x % kSplitLanes;
  const uint32_t w0 = wg * kSplitVw + (local >> 6);
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(w0, a.n_waves - 1));
  const uint32_t vbase = a.waves[w].vbase, count = a.waves[w].count;
  const bool active = (w0 < a.n_waves) && ((local & 63u) < count);
  const uint32_t v = active ? vbase + (local & 63u) : vbase;
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
  float* __restrict__ rows = a.rows + (size_t)wg * 2 * a.frames;
  for (uint32_t t = threadIdx.x; t < 2 * a.frames; t += threads) rows[t] = 0.0f;
  return true;
}
template <int ROLES, int LFO_MODE, bool RETUNE>
__device__ __forceinline__ void welsh_split_dispatch_front(uint32_t cls, UniformArgsPtr ka) {
#define GROOVE_CLS_CASE(CL, C1, C2) case wg_class_combo(CL, C1, C2): welsh_split_front<ROLES, LFO_MODE, RETUNE, C1, C2, CL>(ka); break;
#define GROOVE_CLS_ROW(CL, C1) GROOVE_CLS_CASE(CL, C1, 0) GROOVE_CLS_CASE(CL, C1, 1) GROOVE_CLS_CASE(CL, C1, 2) GROOVE_CLS_CASE(CL, C1, 3) GROOVE_CLS_CASE(CL, C1, 4)
#define GROOVE_CLS_PLANE(CL) GROOVE_CLS_ROW(CL, 0) GROOVE_CLS_ROW(CL, 1) GROOVE_CLS_ROW(CL, 2) GROOVE_CLS_ROW(CL, 3) GROOVE_CLS_ROW(CL, 4)
  switch (cls) {
    GROOVE_CLS_PLANE(OSC_ANY) GROOVE_CLS_PLANE(OSC_TRIANGLE) GROOVE_CLS_PLANE(OSC_SINE)
    default:
      if constexpr (LFO_MODE == LFO_F32) {
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
// One launch for the workgroups of the four class-specialised base kinds of a mid-size bank (the host's workgroup list is
// sorted by kind: they are its first `n_wgs` entries; the exact-f64 kinds behind them take the all-kinds kernel).
#ifdef GROOVE_WELSH_SPLIT_TU /* -DGROOVE_WELSH_SPLIT_TU=3 or =2: one translation unit per number of roles (each carries its own 450 fronts) */
#ifndef GROOVE_WAVES_SPLIT
#if GROOVE_WELSH_SPLIT_TU == 4
#define GROOVE_WAVES_SPLIT 5 /* four roles: 104 VGPRs x 16 wavefronts leave a quarter of every SIMD's registers to the other banks of a project
                                (FM, sampler).  Measured: config #5 0.100 ms per block (0.107 at 128 VGPRs: its other banks wait for the
                                workgroup to leave), 65,536 voices alone 0.0855 either way */
#else
#define GROOVE_WAVES_SPLIT 4 /* 128 VGPRs.  Three roles: one workgroup of twelve wavefronts per CU, one role of each kind per SIMD (at 6 — two
                                workgroups — the smooth-f64-LFO fronts spill 256 bytes per lane; at 5 the three-role form runs 0.1115
                                against 0.0903); two roles: two workgroups of eight */
#endif
#endif
template <bool FUSED, int ROLES>
__global__ __launch_bounds__(ROLES * kSplitLanes, GROOVE_WAVES_SPLIT) GROOVE_NO_TAIL_CALLS void welsh_render_split_kernel(UniformArgs a, const uint8_t* __restrict__ wg_base) {
  const UniformArgsPtr ka = (UniformArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (FUSED) { tp_reduce_prev(a.prev, threadIdx.x, blockIdx.x, gridDim.x); if (welsh_split_idle_workgroup(a, ROLES * kSplitLanes)) return; }
  // (s_setprio 3 here changes nothing: config #5 0.100-0.105 against 0.103-0.107 ms per block, round 3)
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg_base[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wg_cls[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kSplitLanes));
  const bool retune = (base & 1u) != 0;
  if (role == 0) {
    switch (base) {
      case wg_base_kind_of(LFO_F32, false): welsh_split_dispatch_front<ROLES, LFO_F32, false>(cls, ka); break;
      case wg_base_kind_of(LFO_F32, true): welsh_split_dispatch_front<ROLES, LFO_F32, true>(cls, ka); break;
      case wg_base_kind_of(LFO_F64_SMOOTH, false): welsh_split_dispatch_front<ROLES, LFO_F64_SMOOTH, false>(cls, ka); break;
      default: welsh_split_dispatch_front<ROLES, LFO_F64_SMOOTH, true>(cls, ka); break;
    }
  } else if (ROLES == 3 && role == 1) {
    if (retune) welsh_split_mid<true>(ka); else welsh_split_mid<false>(ka);
  } else {
    if (retune) welsh_split_back<ROLES, FUSED, true>(ka); else welsh_split_back<ROLES, FUSED, false>(ka);
  }
}
#if GROOVE_WELSH_SPLIT_TU == 4
template <int LFO_MODE, bool RETUNE>
__device__ __forceinline__ void welsh_split4_dispatch_ctl(uint32_t cl, UniformArgsPtr ka) {
  switch (cl) {
    case OSC_ANY: welsh_split4_ctl<LFO_MODE, RETUNE, OSC_ANY>(ka); break;
    case OSC_TRIANGLE: welsh_split4_ctl<LFO_MODE, RETUNE, OSC_TRIANGLE>(ka); break;
    case OSC_SINE: welsh_split4_ctl<LFO_MODE, RETUNE, OSC_SINE>(ka); break;
    default:
      if constexpr (LFO_MODE == LFO_F32) {
        switch (cl) {
          case OSC_PULSE: welsh_split4_ctl<LFO_MODE, RETUNE, OSC_PULSE>(ka); break;
          case OSC_SAW: welsh_split4_ctl<LFO_MODE, RETUNE, OSC_SAW>(ka); break;
          default: welsh_split4_ctl<LFO_MODE, RETUNE, LFO_UNUSED>(ka); break;
        }
      }
      break;
  }
}
template <int LFO_MODE>
__device__ __forceinline__ void welsh_split4_dispatch_osc(uint32_t c12, UniformArgsPtr ka) {
#define GROOVE_OSC_CASE(C1, C2) case C1 * OSC_CLASSES + C2: welsh_split4_osc<LFO_MODE, C1, C2>(ka); break;
#define GROOVE_OSC_ROW(C1) GROOVE_OSC_CASE(C1, 0) GROOVE_OSC_CASE(C1, 1) GROOVE_OSC_CASE(C1, 2) GROOVE_OSC_CASE(C1, 3) GROOVE_OSC_CASE(C1, 4)
  switch (c12) {
    GROOVE_OSC_ROW(0) GROOVE_OSC_ROW(1) GROOVE_OSC_ROW(2) GROOVE_OSC_ROW(3) GROOVE_OSC_ROW(4)
    default: break;
  }
#undef GROOVE_OSC_ROW
#undef GROOVE_OSC_CASE
}
template <bool FUSED>
__global__ __launch_bounds__(4 * kSplitLanes, GROOVE_WAVES_SPLIT) GROOVE_NO_TAIL_CALLS void welsh_render_split4_kernel(UniformArgs a, const uint8_t* __restrict__ wg_base) {
  const UniformArgsPtr ka = (UniformArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (FUSED) { tp_reduce_prev(a.prev, threadIdx.x, blockIdx.x, gridDim.x); if (welsh_split_idle_workgroup(a, 4 * kSplitLanes)) return; }
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg_base[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wg_cls[GROOVE_WG_SLOT(a.n_wgs)]);
  const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kSplitLanes));
  const bool retune = (base & 1u) != 0;
  const bool f32 = base == (uint32_t)wg_base_kind_of(LFO_F32, false) || base == (uint32_t)wg_base_kind_of(LFO_F32, true);
  if (role == 0) {
    const uint32_t cl = cls / (OSC_CLASSES * OSC_CLASSES);
    if (f32) { if (retune) welsh_split4_dispatch_ctl<LFO_F32, true>(cl, ka); else welsh_split4_dispatch_ctl<LFO_F32, false>(cl, ka); }
    else { if (retune) welsh_split4_dispatch_ctl<LFO_F64_SMOOTH, true>(cl, ka); else welsh_split4_dispatch_ctl<LFO_F64_SMOOTH, false>(cl, ka); }
  } else if (role == 1) {
    const uint32_t c12 = cls % (OSC_CLASSES * OSC_CLASSES);
    if (f32) welsh_split4_dispatch_osc<LFO_F32>(c12, ka); else welsh_split4_dispatch_osc<LFO_F64_SMOOTH>(c12, ka);
  } else if (role == 2) {
    if (retune) welsh_split4_mid<true>(ka); else welsh_split4_mid<false>(ka);
  } else {
    if (retune) welsh_split4_back<FUSED, true>(ka); else welsh_split4_back<FUSED, false>(ka);
  }
}
#endif
#endif
void launch_welsh_split4(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done = nullptr); // four roles:  csrc/welsh_split.hip -DGROOVE_WELSH_SPLIT_TU=4
void launch_welsh_split(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done = nullptr);  // three roles: csrc/welsh_split.hip -DGROOVE_WELSH_SPLIT_TU=3
void launch_welsh_split2(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done = nullptr); // two roles:   csrc/welsh_split.hip -DGROOVE_WELSH_SPLIT_TU=2

} // namespace groove
