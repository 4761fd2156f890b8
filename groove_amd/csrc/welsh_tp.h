// welsh_tp.h — time-parallel evaluation of a Welsh voice: ONE WAVEFRONT PER VOICE, lanes = time.
//
// Why.  A voice's frame loop is a recurrence, so the render kernels in kernels.h give a voice to one lane
// and walk the block's frames serially: a block costs what ONE wavefront needs for 256 dependent frames
// (about 0.11 ms), however few voices the bank has.  Every bank below ~250,000 voices sits on that floor
// (BASELINE configs #2, #3, #5; every synth of a real project; every shard of a strong-scaled run).
//
// What is actually sequential in a voice is small:
//   * oscillator / LFO phases are sums of per-frame increments: a closed form (constant increments) or a
//     prefix sum over frames (pitch LFO); hard sync is "phase 2 restarts at oscillator 1's last wrap": a
//     max-scan of wrap positions plus a difference of two prefix sums;
//   * envelopes are closed forms in a stage counter; stages change at most a handful of times per block,
//     so the state at any frame is reached by stepping over whole stages (env_seek);
//   * the 24 dB filter is LINEAR in its four state values: over a run of frames it is an affine map
//     s -> Phi s + z, with Phi and z depending only on that run's coefficients and input.  Affine maps
//     compose associatively, so the 64 lanes each build the map of their own 4 frames (five short
//     recurrences: the forced response from zero state and the responses to the four unit states), a
//     log-step scan over the lanes gives every lane its true start state, and a second 4-frame pass
//     produces the outputs.  All in f64, like the serial kernels' recurrence.
//   * only the integer noise generator has no jump-ahead (xor and add mixed): a noise oscillator's values are
//     produced serially by one lane first (3 integer operations per tick) and handed out through LDS.
// A block of a voice is then ~1,500 instructions deep instead of ~25,000, at ~4x the total work: the right
// trade exactly where the serial form leaves the machine idle.  groove_hip.hip picks this kernel for Welsh
// banks of up to kTpMaxVoices voices and blocks of up to 256 frames; results agree with the serial kernels to
// f64 rounding of the filter (1e-13 relative) and with the oracle to the same tolerance as they do.
//
// The per-lane pieces are plain inline functions (host + device), so tests/emul runs the same text on the
// CPU with a loop in place of the wavefront.
#pragma once
#include "dsp_core.h"

namespace groove {

constexpr uint32_t kTpLanes = 64, kTpChunk = 4, kTpMaxFrames = kTpLanes * kTpChunk;

// ------------------------------------------------------------------ envelopes at an arbitrary frame
// k more ticks of an envelope, stage by stage: counter and stage exactly as k calls of env_tick leave them
// (`value` is refreshed by the next env_tick, or by env_last_value).
GROOVE_HD void env_seek(EnvState& s, const EnvParams& p, uint32_t k) {
  while (k > 0) {
    env_boundary(s, p);
    if (s.n >= s.N) { s.n += 1; k -= 1; continue; } // more than two chained zero-length stages: the tick advances anyway
    const uint32_t room = s.N - s.n;
    const uint32_t step = k < room ? k : room;
    s.n += step; k -= step;
  }
}
// First frame of the next `frames` at which the amp envelope is idle after its tick (the voice is silent
// from there on: notes land at block starts only); `frames` if it sounds throughout.
GROOVE_HD uint32_t env_idle_at(EnvState s, const EnvParams& p, uint32_t frames) {
  uint32_t done = 0;
  while (done < frames) {
    env_boundary(s, p);
    if (s.state == ENV_IDLE) return done;
    if (s.n >= s.N) { s.n += 1; done += 1; continue; }
    const uint32_t room = s.N - s.n, left = frames - done;
    const uint32_t step = left < room ? left : room;
    s.n += step; done += step;
  }
  return frames;
}

// ------------------------------------------------------------------ the filter as an affine map
// Phi's columns (the section-1 state never sees section 2, so columns 2 and 3 have only their lower halves)
// and the forced response z.
struct Lp24Affine { double c0[4], c1[4], c2[2], c3[2], z[4]; };
GROOVE_HD void lp24_affine_identity(Lp24Affine& m) {
  m.c0[0] = 1.0; m.c0[1] = 0.0; m.c0[2] = 0.0; m.c0[3] = 0.0;
  m.c1[0] = 0.0; m.c1[1] = 1.0; m.c1[2] = 0.0; m.c1[3] = 0.0;
  m.c2[0] = 1.0; m.c2[1] = 0.0; m.c3[0] = 0.0; m.c3[1] = 1.0;
  m.z[0] = 0.0; m.z[1] = 0.0; m.z[2] = 0.0; m.z[3] = 0.0;
}
// one filter step with input x on a plain state vector (the operations of lp24_step, in its order)
GROOVE_HD double lp24_step_v(double* s, const Lp24CoefD& c, double x) {
  const double bx = c.b0a * x;
  const double y1 = bx + s[0];
  s[0] = fma(c.a1a, y1, 2.0 * bx + s[1]);
  s[1] = fma(c.a2a, y1, bx);
  const double by = c.b0b * y1;
  const double y2 = by + s[2];
  s[2] = fma(c.a1b, y2, 2.0 * by + s[3]);
  s[3] = fma(c.a2b, y2, by);
  return y2;
}
GROOVE_HD void lp24_step_h(double* s, const Lp24CoefD& c) { // homogeneous (x = 0)
  const double y1 = s[0];
  s[0] = fma(c.a1a, y1, s[1]);
  s[1] = c.a2a * y1;
  const double by = c.b0b * y1;
  const double y2 = by + s[2];
  s[2] = fma(c.a1b, y2, 2.0 * by + s[3]);
  s[3] = fma(c.a2b, y2, by);
}
GROOVE_HD void lp24_step_h2(double* s23, const Lp24CoefD& c) { // homogeneous, section 2 alone
  const double y2 = s23[0];
  s23[0] = fma(c.a1b, y2, s23[1]);
  s23[1] = c.a2b * y2;
}
// m <- (one more frame with coefficients c and input x) o m
GROOVE_HD void lp24_affine_push(Lp24Affine& m, const Lp24CoefD& c, double x) {
  lp24_step_v(m.z, c, x);
  lp24_step_h(m.c0, c);
  lp24_step_h(m.c1, c);
  lp24_step_h2(m.c2, c);
  lp24_step_h2(m.c3, c);
}
// Phi v + (with_z ? z : 0)
GROOVE_HD void lp24_affine_mul(const Lp24Affine& m, const double* v, double* out, bool with_z) {
  const double z0 = with_z ? m.z[0] : 0.0, z1 = with_z ? m.z[1] : 0.0, z2 = with_z ? m.z[2] : 0.0, z3 = with_z ? m.z[3] : 0.0;
  out[0] = fma(m.c0[0], v[0], fma(m.c1[0], v[1], z0));
  out[1] = fma(m.c0[1], v[0], fma(m.c1[1], v[1], z1));
  out[2] = fma(m.c0[2], v[0], fma(m.c1[2], v[1], fma(m.c2[0], v[2], fma(m.c3[0], v[3], z2))));
  out[3] = fma(m.c0[3], v[0], fma(m.c1[3], v[1], fma(m.c2[1], v[2], fma(m.c3[1], v[3], z3))));
}
// later <- later o earlier   (apply `earlier` first)
GROOVE_HD void lp24_affine_compose(Lp24Affine& later, const Lp24Affine& earlier) {
  Lp24Affine r;
  lp24_affine_mul(later, earlier.c0, r.c0, false);
  lp24_affine_mul(later, earlier.c1, r.c1, false);
  const double e2[4] = {0.0, 0.0, earlier.c2[0], earlier.c2[1]}, e3[4] = {0.0, 0.0, earlier.c3[0], earlier.c3[1]};
  double t[4];
  lp24_affine_mul(later, e2, t, false); r.c2[0] = t[2]; r.c2[1] = t[3];
  lp24_affine_mul(later, e3, t, false); r.c3[0] = t[2]; r.c3[1] = t[3];
  lp24_affine_mul(later, earlier.z, r.z, true);
  later = r;
}

// ------------------------------------------------------------------ one frame of feed-forward work
// Per-frame phase increments of the two audio oscillators (pass 1 of the kinds whose phases need a scan:
// pitch-routed LFO, hard sync).  lfo_phase is the LFO's phase AT this frame.
GROOVE_HD void welsh_tp_incs(const WelshParams& p, const WelshState& s0, uint64_t lfo_phase, float nzl, uint64_t& inc1, uint64_t& inc2) {
  inc1 = s0.o1_inc; inc2 = s0.o2_inc;
  if (p.flags & WF_LFO_PITCH) {
    const uint32_t wl = (p.flags >> WF_LFO_WAVE_SHIFT) & 15u;
    const double l = osc_value_f64(wl, lfo_phase, 0x8000000000000000ull, nzl);
    const double m = exp2_small_f64(l * (double)p.lfo_depth);
    if (p.flags & WF_LFO_O1) inc1 = f64_to_u64((double)inc1 * m);
    if (p.flags & WF_LFO_O2) inc2 = f64_to_u64((double)inc2 * m);
  }
}
// Everything of a LIVE frame except the filter: the mixed oscillator sample x (the filter's input), the
// amplitude factor a (envelope x amplitude LFO), and — retuned kinds — this frame's filter coefficients.
// `s` is the lane's running state: both envelopes have been ticked for this frame by the caller; the LFO
// phase and (constant-increment kinds) the oscillator phases are advanced here exactly as welsh_frame does;
// kinds with scanned phases pass this frame's phases in (ext_phase).
// RESO: the resonance routing is compiled in.  t_out / hi_out: the tangent (and its side of SR/4) behind the coefficients, set
// whenever they are recomputed from a cutoff percent (lp24_coefd_from_t(p.fc, t_out, hi_out) == coef, bit for bit).
template <bool RETUNE, bool RESO = true>
GROOVE_HD void welsh_tp_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, bool is_first, bool ext_phase,
                              uint64_t ph1, uint64_t ph2, float nz1, float nz2, float nzl,
                              Lp24CoefD& coef, float& prev_pct, float& x, float& a, float& t_out, bool& hi_out) {
  const uint32_t fl = p.flags;
  const uint32_t w1 = (fl >> WF_O1_WAVE_SHIFT) & 15u, w2 = (fl >> WF_O2_WAVE_SHIFT) & 15u, wl = (fl >> WF_LFO_WAVE_SHIFT) & 15u;
  const uint64_t half = 0x8000000000000000ull;
  if (!is_first) s.lfo.phase += p.lfo_inc;
  uint64_t d1 = p.o1_duty64, d2 = p.o2_duty64;
  float lfo = 0.0f;
  if (fl & WF_LFO_PW) {
    const double ld = osc_value_f64(wl, s.lfo.phase, half, nzl) * (double)p.lfo_depth;
    if (fl & WF_LFO_O1) d1 = f64_to_u64(clamp01d((double)p.o1_duty * (1.0 + ld)) * 18446744073709549568.0);
    if (fl & WF_LFO_O2) d2 = f64_to_u64(clamp01d((double)p.o2_duty * (1.0 + ld)) * 18446744073709549568.0);
  } else if (fl & (WF_LFO_AMP | WF_LFO_CUTOFF | WF_LFO_RESO)) {
    lfo = osc_value(wl, s.lfo.phase, half, nzl);
  }
  if (ext_phase) {
    s.o1.phase = ph1; s.o2.phase = ph2;
  } else if (!is_first) {
    s.o1.phase += s.o1_inc; s.o2.phase += s.o2_inc;
  }
  const float v1 = osc_value(w1, s.o1.phase, d1, nz1);
  const float v2 = osc_value(w2, s.o2.phase, d2, nz2);
  x = fmaf(v1, p.mix, v2 * (1.0f - p.mix));
  if (RETUNE) {
    bool retune = false;
    float pct = 0.0f;
    if (fl & WF_RETUNE_ENV) {
      pct = fmaf((1.0f - p.cutoff_start) * p.cutoff_end, s.fil.value, p.cutoff_start);
      retune = true;
    } else if (fl & WF_LFO_CUTOFF) {
      pct = p.cutoff_start * fmaf(lfo, p.lfo_depth, 1.0f);
      retune = true;
    }
    if (RESO && (fl & WF_LFO_RESO)) {
      const Lp24Consts c = lp24_consts_from_ripple(p.ripple * fmaf(lfo, p.lfo_depth, 1.0f));
      const float fc = retune ? 25.0f * fast_exp2(clamp01f(pct) * 9.6438561897747244f) : p.cutoff_hz;
      coef = lp24_coefd_from_fc(c, fc, rc.pi_over_sr, rc.fc_max);
    } else if (retune && pct != prev_pct) {
      t_out = lp24_t_from_pct(pct, rc, hi_out); // (lp24_coefd_from_pct in its two halves)
      // (WF_COEF_WIDE: the two-sided form of the serial kernels' retune, round 6 — lp24_coefd_from_fc's before, which a drawn patch with a
      // pulse LFO on the cutoff at ripple 4.0 and 22,050 Hz played 1.3e-5 of its level off the oracle where the serial forms stood at 1.1e-6)
      coef = lp24_coefd_from_t(p.fc, t_out, hi_out, (fl & WF_COEF_WIDE) != 0);
      prev_pct = pct;
    }
  }
  a = s.amp.value;
  if (fl & WF_LFO_AMP) a *= fmaf(lfo, p.lfo_depth, 1.0f);
}
// What pass B needs of a lane's frames: the filter's input, the amplitude factor, and — retuned kinds — the tangent the frame's
// coefficients are made from (lp24_coefd_from_t; bit j of `hi` = its side of SR/4).  Recomputing six f64 coefficients from one
// float costs ~25 instructions per frame and keeps CH x 12 registers free across the affine scan (the kernel's register peak:
// three 32-register maps are alive there).
template <int CH> struct TpChunkOut { float x[CH], amp[CH], t[CH]; uint32_t hi, live; };
// Pass 2 of a lane: frames n0 .. n0 + cnt - 1 (the first `nlive` of them sound).  `s` arrives at frame n0 (envelopes sought,
// phases advanced); it leaves as the serial walk leaves it after the lane's last frame.  FULL_COEF: the resonance routing's
// coefficients do not come from a tangent alone; such banks keep the six f64 coefficients per frame (coef_full).
// (A stage-by-stage form — every oscillator's waveform dispatched once per lane instead of once per frame — was built and
// measured in round 3: same time, 20 more registers.  The frames are walked one by one.)
template <bool RETUNE, bool FULL_COEF, int CH>
GROOVE_HD void welsh_tp_chunk(const WelshParams& p, WelshState& s, const RenderConsts& rc, bool first0, uint32_t n0, uint32_t cnt, uint32_t nlive,
                              bool ext_phase, const uint64_t (&ph1)[CH], const uint64_t (&ph2)[CH],
                              const float* nz1, const float* nz2, const float* nzl /* the voice's noise rows (by frame), or null */,
                              const Lp24CoefD& cur0, TpChunkOut<CH>& o, Lp24CoefD (&coef_full)[FULL_COEF ? CH : 1], Lp24Affine& mine) {
  Lp24CoefD cur = cur0;
  float prev_pct = __builtin_nanf(""), t_cur = 0.0f;
  bool hi_cur = false;
  o.live = 0; o.hi = 0;
  _Pragma("unroll") for (uint32_t j = 0; j < (uint32_t)CH; ++j) {
    o.x[j] = 0.0f; o.amp[j] = 0.0f;
    if (j < cnt) {
      env_tick(s.amp, p.amp);
      env_tick(s.fil, p.fil);
      if (j < nlive) {
        o.live |= 1u << j;
        const bool is_first = first0 && n0 + j == 0;
        const uint32_t f = n0 + j;
        welsh_tp_frame<RETUNE, FULL_COEF>(p, s, rc, is_first, ext_phase, ph1[j], ph2[j], nz1 ? nz1[f] : 0.0f, nz2 ? nz2[f] : 0.0f, nzl ? nzl[f] : 0.0f,
                                          cur, prev_pct, o.x[j], o.amp[j], t_cur, hi_cur);
        s.vflags = 0;
        lp24_affine_push(mine, cur, (double)o.x[j]);
      } else if (ext_phase) { // phases stay where the last live frame left them
        s.o1.phase = ph1[j]; s.o2.phase = ph2[j];
      }
    }
    o.t[j] = t_cur;
    if (hi_cur) o.hi |= 1u << j;
    if (FULL_COEF) coef_full[FULL_COEF ? j : 0] = cur;
  }
}
// The coefficients pass B applies at frame j of the lane: what pass 2 pushed into the affine map, bit for bit.
template <bool RETUNE, bool FULL_COEF, int CH>
GROOVE_HD Lp24CoefD welsh_tp_coef_at(const WelshParams& p, const TpChunkOut<CH>& o, const Lp24CoefD& cur0, const Lp24CoefD (&coef_full)[FULL_COEF ? CH : 1], uint32_t j) {
  if (FULL_COEF) return coef_full[FULL_COEF ? j : 0];
  if (RETUNE && (p.flags & (WF_RETUNE_ENV | WF_LFO_CUTOFF))) return lp24_coefd_from_t(p.fc, o.t[j], ((o.hi >> j) & 1u) != 0, (p.flags & WF_COEF_WIDE) != 0);
  return cur0;
}
GROOVE_HD bool welsh_tp_scans(const WelshParams& p) { return (p.flags & (WF_LFO_PITCH | WF_SYNC)) != 0; }
GROOVE_HD bool welsh_tp_noise(const WelshParams& p, int osc /*0: osc 1, 1: osc 2, 2: LFO*/) {
  const uint32_t sh = osc == 0 ? (uint32_t)WF_O1_WAVE_SHIFT : (osc == 1 ? (uint32_t)WF_O2_WAVE_SHIFT : (uint32_t)WF_LFO_WAVE_SHIFT);
  return ((p.flags >> sh) & 15u) == GROOVE_WAVE_NOISE;
}

// ------------------------------------------------------------------ FmVoice, time-parallel
// An FM voice has no filter: the modulator's phase is a closed form, the carrier's a prefix sum of per-frame
// (signed, through-zero) increments.  One frame's carrier increment, given the modulator phase AT that frame and
// the modulator envelope's value (fm_frame's arithmetic):
GROOVE_HD uint64_t fm_tp_carrier_inc(const FmParams& p, uint64_t c_inc, uint64_t mod_phase, float menv_value) {
  const double mv = osc_value_f64(GROOVE_WAVE_SINE, mod_phase, 0, 0.0f);
  const double lfm = mv * (double)menv_value * p.depth_beta;
  return turns_to_inc((double)c_inc * 5.42101086242752217004e-20 * (1.0 + lfm));
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ the kernel
#ifndef GROOVE_TP_WAVES
#define GROOVE_TP_WAVES 3
#endif
constexpr int kTpWaves = 4;                 // voices per workgroup
constexpr int kTpThreads = kTpWaves * 64;
// Voice group of workgroup `i` of a grid of `grid` workgroups (welsh_tp_kernel, "XCD-aware voice mapping"); the host pads
// the grid of a bank of 16 or more groups to a multiple of 8 (welsh_tp_grid), smaller grids keep the plain order.
__host__ __device__ inline uint32_t tp_group_of_block(uint32_t i, uint32_t grid) {
  return ((grid & 7u) == 0 && grid >= 16u) ? (i & 7u) * (grid >> 3) + (i >> 3) : i;
}
// sampler voices are a gather and nothing else: sixteen to a workgroup, so that the bus reduction behind the render has
// a quarter of the partial rows to read (sampler-16384: 4,096 rows of 2 KB -> 1,024)
constexpr int kSamplerTpWaves = 16;
constexpr int kSamplerTpThreads = kSamplerTpWaves * 64;
constexpr uint32_t kTpMaxVoices = 16384;    // measured crossover with the serial kernels ~24,000 voices (tools/tp_bench.py)

// N ticks of the noise generator on the scalar unit (wave-uniform state x1, x2): the value of tick I — x2 before the add —
// goes to lane I of `acc` (v_writelane_b32 with the lane as an inline constant)
template <int N, int I = 0> __device__ __forceinline__ void tp_noise_ticks(uint32_t& x1, uint32_t& x2, int& acc) {
  if constexpr (I < N) {
    x1 ^= x2;
    asm("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(x2), "i"(I));
    x2 += x1;
    tp_noise_ticks<N, I + 1>(x1, x2, acc);
  }
}
template <class T> __device__ __forceinline__ T tp_shfl(T x, int src) {
  static_assert(sizeof(T) == 8 || sizeof(T) == 4, "");
  if constexpr (sizeof(T) == 8) {
    const uint64_t b = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(b >> 32), src, 64);
    return __builtin_bit_cast(T, ((uint64_t)hi << 32) | lo);
  } else {
    return __builtin_bit_cast(T, __shfl(__builtin_bit_cast(int, x), src, 64));
  }
}
__device__ __forceinline__ void tp_shfl_affine(const Lp24Affine& m, int src, Lp24Affine& out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { out.c0[i] = tp_shfl(m.c0[i], src); out.c1[i] = tp_shfl(m.c1[i], src); out.z[i] = tp_shfl(m.z[i], src); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { out.c2[i] = tp_shfl(m.c2[i], src); out.c3[i] = tp_shfl(m.c3[i], src); }
}
// Inclusive scan of the affine maps over each group of LPV lanes, by DPP (round 3).  Rows of 16 lanes first (row_shr 1, 2, 4, 8),
// then the rows' totals into the rows behind them (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 — the second
// only when a voice spans the wavefront).  A lane with nothing to receive gets the IDENTITY map from the instruction itself (the
// `old` operand of a DPP move whose source is out of range or whose row is masked), and composing with the identity is exact, so
// there is no select; and a DPP move is a register move, where ds_bpermute is an LDS round trip a lone wavefront cannot hide:
// six steps of 32 ds_bpermute + 32 v_cndmask + 44 f64 operations became 32 DPP moves + 44 f64 operations each.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double tp_dpp_f64(double old, double x) {
  const uint64_t o = __builtin_bit_cast(uint64_t, old), b = __builtin_bit_cast(uint64_t, x);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tp_dpp_compose(Lp24Affine& incl) { // incl <- incl o (the map CTRL brings, or the identity)
  Lp24Affine e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e.c0[i] = tp_dpp_f64<CTRL, ROW_MASK>(i == 0 ? 1.0 : 0.0, incl.c0[i]);
    e.c1[i] = tp_dpp_f64<CTRL, ROW_MASK>(i == 1 ? 1.0 : 0.0, incl.c1[i]);
    e.z[i] = tp_dpp_f64<CTRL, ROW_MASK>(0.0, incl.z[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    e.c2[i] = tp_dpp_f64<CTRL, ROW_MASK>(i == 0 ? 1.0 : 0.0, incl.c2[i]);
    e.c3[i] = tp_dpp_f64<CTRL, ROW_MASK>(i == 1 ? 1.0 : 0.0, incl.c3[i]);
  }
  lp24_affine_compose(incl, e);
}
template <int LPV>
__device__ __forceinline__ void tp_scan_affine(Lp24Affine& incl) {
  static_assert(LPV == 64 || LPV == 32, "");
  tp_dpp_compose<0x111, 0xF>(incl); // row_shr:1
  tp_dpp_compose<0x112, 0xF>(incl); // row_shr:2
  tp_dpp_compose<0x114, 0xF>(incl); // row_shr:4
  tp_dpp_compose<0x118, 0xF>(incl); // row_shr:8
  tp_dpp_compose<0x142, 0xA>(incl); // row_bcast:15 -> rows 1 and 3
  if constexpr (LPV == 64) tp_dpp_compose<0x143, 0xC>(incl); // row_bcast:31 -> rows 2 and 3
}
// the same pattern for the prefix sums of the two oscillators' increments (identity 0) and the running maximum of the wrap
// positions (identity -1: "none")
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t tp_dpp_u64(uint64_t old, uint64_t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)x, CTRL, ROW_MASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(x >> 32), CTRL, ROW_MASK, 0xF, false);
  return ((uint64_t)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tp_dpp_add2(uint64_t& a, uint64_t& b) { a += tp_dpp_u64<CTRL, ROW_MASK>(0, a); b += tp_dpp_u64<CTRL, ROW_MASK>(0, b); }
template <int LPV>
__device__ __forceinline__ void tp_scan_add_u64(uint64_t& a, uint64_t& b) {
  tp_dpp_add2<0x111, 0xF>(a, b); tp_dpp_add2<0x112, 0xF>(a, b); tp_dpp_add2<0x114, 0xF>(a, b); tp_dpp_add2<0x118, 0xF>(a, b);
  tp_dpp_add2<0x142, 0xA>(a, b);
  if constexpr (LPV == 64) tp_dpp_add2<0x143, 0xC>(a, b);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tp_dpp_max(int& x) { const int o = __builtin_amdgcn_update_dpp(-1, x, CTRL, ROW_MASK, 0xF, false); x = o > x ? o : x; }
template <int LPV>
__device__ __forceinline__ void tp_scan_max_i32(int& x) {
  tp_dpp_max<0x111, 0xF>(x); tp_dpp_max<0x112, 0xF>(x); tp_dpp_max<0x114, 0xF>(x); tp_dpp_max<0x118, 0xF>(x);
  tp_dpp_max<0x142, 0xA>(x);
  if constexpr (LPV == 64) tp_dpp_max<0x143, 0xC>(x);
}
// y-recurrence of the Direct Form 1 biquad as an affine map of (y1, y2): y = w - a1 y1 - a2 y2, w = b0 x + b1 x1 + b2 x2
struct BqAffine { double m00, m01, m10, m11, z0, z1; }; // (y1, y2)' = M (y1, y2) + z
__device__ __forceinline__ void bq_affine_identity(BqAffine& m) { m.m00 = 1.0; m.m01 = 0.0; m.m10 = 0.0; m.m11 = 1.0; m.z0 = 0.0; m.z1 = 0.0; }
__device__ __forceinline__ void bq_affine_push(BqAffine& m, double a1, double a2, double w) {
  // one more frame: (y1, y2) -> (w - a1 y1 - a2 y2, y1)
  const double n00 = -a1 * m.m00 - a2 * m.m10, n01 = -a1 * m.m01 - a2 * m.m11, nz0 = (w - a1 * m.z0) - a2 * m.z1;
  m.m10 = m.m00; m.m11 = m.m01; m.z1 = m.z0;
  m.m00 = n00; m.m01 = n01; m.z0 = nz0;
}
__device__ __forceinline__ void bq_affine_compose(BqAffine& later, const BqAffine& e) { // later <- later o e
  BqAffine r;
  r.m00 = later.m00 * e.m00 + later.m01 * e.m10; r.m01 = later.m00 * e.m01 + later.m01 * e.m11;
  r.m10 = later.m10 * e.m00 + later.m11 * e.m10; r.m11 = later.m10 * e.m01 + later.m11 * e.m11;
  r.z0 = later.m00 * e.z0 + later.m01 * e.z1 + later.z0;
  r.z1 = later.m10 * e.z0 + later.m11 * e.z1 + later.z1;
  later = r;
}

// One lane-channel's block on one wavefront: lane l holds frames 4l .. 4l+3 (xf; `cnt` of them exist).  o = the outputs of
// this lane's frames (wet mix applied); returns true in the lane that holds the block's last frame, with ns = the state
// after the block (x1, x2 = the last two inputs, y1, y2 = the last two outputs).  Pass B performs biquad_step's operations
// in their order from a start state that agrees with the serial walk's to f64 rounding.
template <int CH = (int)kTpChunk, int LPV = 64> // LPV lanes of the wavefront hold one lane-channel, CH frames each (welsh_tp_kernel's VPW = 64 / LPV)
__device__ __forceinline__ bool bq_tp_wave(const float (&xf)[CH], uint32_t cnt, uint32_t lane, uint32_t frames, const BiquadCoefD& c,
                                           double sx1, double sx2, double sy1, double sy2, float wm, float (&o)[CH], double (&ns)[4]) {
  const uint32_t vl = lane & (uint32_t)(LPV - 1);
  // the two inputs before this lane's first frame: the previous lane's last two, or the state
  double px1 = tp_shfl((double)xf[CH - 1], (int)lane - 1), px2 = tp_shfl((double)xf[CH - 2], (int)lane - 1);
  if (vl == 0) { px1 = sx1; px2 = sx2; }
  double w[CH];
  BqAffine mine;
  bq_affine_identity(mine);
  {
    double x1 = px1, x2 = px2;
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      const double x = (double)xf[j];
      w[j] = c.b0 * x + c.b1 * x1 + c.b2 * x2;
      if (j < cnt) bq_affine_push(mine, c.a1, c.a2, w[j]);
      x2 = x1; x1 = x;
    }
  }
  BqAffine incl = mine;
#pragma unroll
  for (int d = 1; d < LPV; d <<= 1) {
    BqAffine e;
    e.m00 = tp_shfl(incl.m00, (int)lane - d); e.m01 = tp_shfl(incl.m01, (int)lane - d);
    e.m10 = tp_shfl(incl.m10, (int)lane - d); e.m11 = tp_shfl(incl.m11, (int)lane - d);
    e.z0 = tp_shfl(incl.z0, (int)lane - d); e.z1 = tp_shfl(incl.z1, (int)lane - d);
    if ((int)vl >= d) bq_affine_compose(incl, e);
  }
  const double e1 = incl.m00 * sy1 + incl.m01 * sy2 + incl.z0, e2 = incl.m10 * sy1 + incl.m11 * sy2 + incl.z1; // (y1, y2) after this lane
  double y1 = tp_shfl(e1, (int)lane - 1), y2 = tp_shfl(e2, (int)lane - 1);
  if (vl == 0) { y1 = sy1; y2 = sy2; }
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    o[j] = 0.0f;
    if (j < cnt) {
      const double y = w[j] - c.a1 * y1 - c.a2 * y2;
      y2 = y1; y1 = y;
      o[j] = (float)y;
      if (wm < 1.0f) o[j] = fmaf(o[j], wm, xf[j] * (1.0f - wm));
    }
  }
  const bool holds_end = frames && vl == (frames - 1) / CH;
  if (holds_end) { // x1, x2 = the block's last two inputs; y1, y2 = its last two outputs
    ns[0] = (double)xf[cnt - 1]; ns[1] = cnt >= 2 ? (double)xf[cnt - 2] : px1; ns[2] = y1; ns[3] = y2;
  }
  return holds_end;
}
struct TpArgs {
  const uint32_t* params; uint32_t* state; float* out; float* rows; size_t ch_stride; RenderConsts rc; uint32_t n, frames;
  TpPrev prev; // (fused form only)
  // welsh_tp_kernel<false, true>: a 12 dB BiQuad effect bank (one lane per voice) applied to the voice's block before it is
  // stored — coefficients [5][n] f64, state [4][2n] f64, wet [n], the layouts of fx_biquad_tp_kernel (fx_tp.h)
  const double* bq_coef = nullptr; double* bq_st = nullptr; const float* bq_wet = nullptr;
  bool full_coef = false; // some voice of the bank has the resonance routing: the kernel form that keeps whole coefficient sets
  uint32_t vpw = 1;       // voices per wavefront (welsh_tp_kernel's VPW; 2 needs every pair of adjacent voices on one patch)
};
// A block's note events, small enough to ride in the kernel's argument block (4 KB): strictly increasing voices, so a
// wavefront finds its voice's event (there is at most one) by bisection with scalar loads and applies it to the state
// it has just read — no separate events kernel, and no host-to-device copy ordered in front of the render.  A sampler
// project with staggered note-ons (config #4) has a few hundred events per block, every block.
constexpr uint32_t kInlineEvents = 440;
struct InlineEvents { uint32_t n; groove_note_event ev[kInlineEvents]; };
// rows = partial[workgroup][ch][frame] (the bus reduction's rows), written by both forms; !FUSED also stores the planar
// block out[ch][frame][voice] (kernels.h run_frames: the rows are then the block's own lane sums for groove_mix).
// HEAD_BQ (block-writing form only): the voice's (L, R) block goes through a BiQuad effect lane before it is stored — the
// effect chain's leading IIR stage, fused (groove_bank_render_chain_async): the 256 frames are already spread over the
// wavefront's lanes, so the filter is one more affine scan (bq_tp_wave, what fx_biquad_tp_kernel runs per lane-channel) on
// values that are in registers; a separate launch read and wrote the whole block again and, as a few hundred latency-bound
// wavefronts beside the HBM-bound stages of the previous block, took three times its own time (config #3: 29 us of 82).
#ifdef GROOVE_TP_PROBE /* measurement build only (tools/tp_probe.py): s_memtime ticks a wavefront spends in each phase of the kernel */
static __device__ unsigned long long g_tp_probe[16]; // [phase 0..11] ticks, [12] realtime ticks of the whole kernel, [15] wavefronts
#define TP_PROBE_BEGIN uint64_t pb_t[12]; const uint64_t pb_r0 = __builtin_amdgcn_s_memrealtime(); int pb_i = 0; TP_PROBE
#define TP_PROBE { __builtin_amdgcn_sched_barrier(0); pb_t[pb_i++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define TP_PROBE_END { TP_PROBE const uint64_t pb_r1 = __builtin_amdgcn_s_memrealtime(); if ((threadIdx.x & 63u) == 0) { \
    for (int i = 1; i < pb_i; ++i) atomicAdd(&g_tp_probe[i - 1], pb_t[i] - pb_t[i - 1]); \
    atomicAdd(&g_tp_probe[12], pb_r1 - pb_r0); atomicAdd(&g_tp_probe[15], 1ull); } }
#else
#define TP_PROBE_BEGIN
#define TP_PROBE
#define TP_PROBE_END
#endif
// VPW = voices per wavefront (round 3).  VPW = 1: lanes = the 64 four-frame chunks of one voice.  VPW = 2: the wavefront's
// halves take two ADJACENT voices of one patch (the host checks that every pair shares its parameter words, so the patch
// stays in SGPRs), 32 lanes x 8 frames each: half the wavefronts, five scan steps instead of six, and the fixed costs of a
// wavefront (parameters, envelope seeks, the scan, the tile turn) paid once per two voices — ~3,500 instructions deep per
// wavefront instead of ~2,350, so 25 % less issue per voice and, where VPW = 1 needs more wavefronts than the SIMDs hold at
// once (165 registers: three per SIMD, i.e. banks over 3,072 voices), a shorter block.  Same per-frame operations: a voice's
// output differs from the VPW = 1 form's only by the f64 rounding of the filter scan's composition order.
// Which workgroup of which launch a time-parallel body is (round 5): `wg` of `n_wg` workgroups of ITS bank (voice mapping), `row` =
// its row pair in rows[][2][frames], and (`pwg`, `pn_wg`) = its index among all workgroups of the launch, which share out the
// previous block's bus reduction (tp_reduce_prev).  A bank's own kernel: all of them blockIdx.x / gridDim.x; the mixed kernel of
// several small banks (tp_mixed_kernel below) gives every bank a range of its grid.
struct TpWg { uint32_t wg, n_wg, row, pwg, pn_wg; };
__device__ __forceinline__ TpWg tp_wg_own() { return TpWg{blockIdx.x, gridDim.x, blockIdx.x, blockIdx.x, gridDim.x}; }
template <int VPW> struct WelshTpSmem {
  float noise[kTpWaves * VPW][3][kTpMaxFrames];
  // hard sync's prefix sums (pass 1) and the output tile (the end) share their 2 KB per voice: until the workgroup barrier in
  // front of the tile turn a wavefront touches its own voices' rows only, and its tile writes follow its last sync gather
  uint64_t sum2[kTpWaves * VPW][kTpMaxFrames];
};
template <bool FUSED, bool HEAD_BQ, bool FULL_COEF, int VPW>
__device__ __forceinline__ void welsh_tp_body(const TpArgs& a, const TpWg g, WelshTpSmem<VPW>& sm) {
  static_assert(!(FUSED && HEAD_BQ), "an effect needs the voice blocks: the fused bus form has none");
  static_assert(VPW == 1 || VPW == 2, "");
  constexpr uint32_t LPV = 64 / VPW, CH = kTpChunk * VPW, WGV = kTpWaves * VPW; // lanes per voice, frames per lane, voices per workgroup
  float (&s_noise)[WGV][3][kTpMaxFrames] = sm.noise;
  uint64_t (&s_sum2)[WGV][kTpMaxFrames] = sm.sum2;
  float (*s_tile)[2][kTpMaxFrames] = reinterpret_cast<float (*)[2][kTpMaxFrames]>(&s_sum2[0][0]);
  static_assert(sizeof(uint64_t) * kTpMaxFrames == sizeof(float) * 2 * kTpMaxFrames, "a voice's tile row is its prefix-sum row");
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t vl = lane & (LPV - 1u), sub = lane / LPV, wv = wave * VPW + sub; // lane within the voice, voice within the wave / workgroup
  const int lbase = (int)(sub * LPV);                                             // the voice's first lane
  TP_PROBE_BEGIN // 0
  // XCD-aware voice mapping.  Workgroups go to the 8 XCDs round-robin (workgroup i -> XCD i mod 8), each XCD with its own L2.
  // A workgroup stores 4 adjacent voices x 256 frames of the planar block [frame][voice]: 16 bytes per row.  With the plain
  // mapping the 8 workgroups that share a 128-byte line of a row sit on 8 DIFFERENT XCDs — eight L2s each hold an eighth
  // of every line and write it back masked; with group = (i mod 8) * (grid / 8) + i / 8 an XCD owns a contiguous range
  // of voices, the workgroups that share a line follow each other on ONE XCD, and its L2 writes whole lines.
  const uint32_t grp = tp_group_of_block(g.wg, g.n_wg);
  const uint32_t v0 = grp * WGV + wv;
  const bool voice = v0 < a.n;
  const uint32_t vme = voice ? v0 : a.n - 1; // (per lane when VPW > 1)
  const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)vme);
  const uint32_t frames = a.frames, n = a.n;
  const WelshParams p = make_scalar(soa_load<WelshParams>(a.params, n, v)); // VPW > 1: the wave's voices share the patch (host)
  const WelshState s0 = soa_load<WelshState>(a.state, n, VPW == 1 ? v : vme); // the same in every lane of the voice
  if constexpr (FUSED) tp_reduce_prev(a.prev, threadIdx.x, g.pwg, g.pn_wg);
  const RenderConsts rc = a.rc;
  const bool first0 = (s0.vflags & VF_FIRST) != 0;
  const uint32_t live_total = env_idle_at(s0.amp, p.amp, frames);
  const bool retunes = welsh_retunes(p), scans = welsh_tp_scans(p);
  const bool nz_any = welsh_tp_noise(p, 0) || welsh_tp_noise(p, 1) || welsh_tp_noise(p, 2);
  TP_PROBE // 1: parameters, state, env_idle_at

  // ---- noise oscillators: the generator has no jump-ahead, so its `live_total` values are produced serially — on the SCALAR
  // unit (round 3): a voice's generator state is wave-uniform, a tick is s_xor + s_add, and the value lands in lane (tick mod 64)
  // of a register by v_writelane; after 64 ticks the 64 lanes convert and store their values together.  Three instructions per
  // tick instead of the eight of a one-lane VALU loop with its LDS store (a noise voice's wavefront was the slowest of its
  // workgroup by ~17,000 cycles per block: the tail of config #2's kernel).  ONE copy of the 64-tick body serves the three
  // oscillators and the wavefront's voices (kernel text is not free here, see pass 2).  End states: lanes 0 .. 2 of the voice.
  OscState nz_end = vl == 0 ? s0.o1 : (vl == 1 ? s0.o2 : s0.lfo);
  if (nz_any) {
#pragma unroll 1
    for (uint32_t job = 0; job < 3u * VPW; ++job) { // (oscillator k, voice sv of the wavefront)
      const uint32_t k = job % 3u, sv = job / 3u;
      if (!welsh_tp_noise(p, (int)k)) continue;
      const uint32_t ox1 = k == 0 ? s0.o1.x1 : (k == 1 ? s0.o2.x1 : s0.lfo.x1), ox2 = k == 0 ? s0.o1.x2 : (k == 1 ? s0.o2.x2 : s0.lfo.x2);
      const int src = (int)(sv * LPV); // a lane of that voice (they all hold its state)
      uint32_t x1 = (uint32_t)__builtin_amdgcn_readlane((int)ox1, src), x2 = (uint32_t)__builtin_amdgcn_readlane((int)ox2, src);
      const uint32_t lt = (uint32_t)__builtin_amdgcn_readlane((int)live_total, src);
      float* row = s_noise[wave * VPW + sv][k];
#pragma unroll 1
      for (uint32_t base = 0; base < lt; base += 64) {
        int acc = 0;
        uint32_t y1 = x1, y2 = x2;
        tp_noise_ticks<64>(y1, y2, acc); // (past lt inside the last 64: values nobody reads; the end state is taken below)
        row[base + lane] = (float)acc * 4.6566128730773926e-10f; // noise_tick's value: x2 before the add, as int32 * 2^-31
        if (base + 64 <= lt) { x1 = y1; x2 = y2; }
        else { for (uint32_t i = base; i < lt; ++i) { x1 ^= x2; x2 += x1; } } // a ragged tail's end state: two scalar operations per tick
      }
      if (sub == sv && vl == k) { nz_end.x1 = x1; nz_end.x2 = x2; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }

  TP_PROBE // 2: noise
  // ---- this lane's frames
  const uint32_t n0 = vl * CH;
  const uint32_t cnt = n0 < frames ? (frames - n0 < CH ? frames - n0 : CH) : 0u;
  WelshState s = s0;
  env_seek(s.amp, p.amp, n0 < frames ? n0 : 0u);
  env_seek(s.fil, p.fil, n0 < frames ? n0 : 0u);
  const uint32_t live_before = n0 < live_total ? n0 : live_total;                   // live frames before n0
  const uint64_t adv = (uint64_t)(live_before - ((first0 && live_before >= 1u) ? 1u : 0u)); // phase advances before n0
  s.lfo.phase = s0.lfo.phase + adv * p.lfo_inc;
  s.o1.phase = s0.o1.phase + adv * s0.o1_inc;
  s.o2.phase = s0.o2.phase + adv * s0.o2_inc;
  if (live_before >= 1u) s.vflags = 0;

  TP_PROBE // 3: env_seek, closed-form phases
  // pass 1 (scanned phases): per-frame increments, prefix sums over the block, wrap positions
  uint64_t ph1[CH] = {}, ph2[CH] = {};
  if (scans) {
    uint64_t inc1[CH], inc2[CH], run1 = 0, run2 = 0, loc1[CH], loc2[CH];
    uint64_t lph = s.lfo.phase;
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      const uint32_t f = n0 + j;
      const bool live = j < cnt && f < live_total, is_first = first0 && f == 0;
      inc1[j] = 0; inc2[j] = 0;
      if (live) {
        if (!is_first) lph += p.lfo_inc;
        const float nzl = welsh_tp_noise(p, 2) ? s_noise[wv][2][f] : 0.0f;
        if (!is_first) welsh_tp_incs(p, s0, lph, nzl, inc1[j], inc2[j]);
      }
      run1 += inc1[j]; run2 += inc2[j];
      loc1[j] = run1; loc2[j] = run2;
    }
    // exclusive prefix over the voice's lanes of the lane totals (Hillis-Steele on the inclusive sums)
    uint64_t t1 = run1, t2 = run2;
    tp_scan_add_u64<LPV>(t1, t2);
    const uint64_t base1 = t1 - run1, base2 = t2 - run2;
    int wlast = -1; // last frame <= f at which oscillator 1 wrapped (hard sync), within this lane so far
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      ph1[j] = s0.o1.phase + base1 + loc1[j];
      if (ph1[j] < inc1[j]) wlast = (int)(n0 + j); // carry out of the add (inc 0: never)
      loc2[j] += base2;                              // inclusive prefix of the oscillator-2 increments at frame n0 + j
    }
    if (p.flags & WF_SYNC) {
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) if (j < cnt) s_sum2[wv][n0 + j] = loc2[j];
      int wl_incl = wlast; // max-scan over the voice's lanes
      tp_scan_max_i32<LPV>(wl_incl);
      int wprev = __shfl(wl_incl, (int)lane - 1, 64);
      if (vl == 0) wprev = -1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      int w = wprev;
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) {
        if (ph1[j] < inc1[j]) w = (int)(n0 + j);
        if (j < cnt) ph2[j] = w >= 0 ? loc2[j] - s_sum2[wv][w] : s0.o2.phase + loc2[j];
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) ph2[j] = s0.o2.phase + loc2[j];
    }
  }

  TP_PROBE // 4: pass 1 (scanned phases)
  // pass 2: the frames' feed-forward values, stage by stage (welsh_tp_chunk); the filter's affine map of this chunk
  const Lp24CoefD cur0 = lp24_coefd_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max);
  const uint32_t nlive = n0 < live_total ? (live_total - n0 < cnt ? live_total - n0 : cnt) : 0u;
  TpChunkOut<CH> co;
  Lp24CoefD coef_full[FULL_COEF ? CH : 1];
  Lp24Affine mine;
  lp24_affine_identity(mine);
  {
    const float* nz1 = welsh_tp_noise(p, 0) ? s_noise[wv][0] : nullptr;
    const float* nz2 = welsh_tp_noise(p, 1) ? s_noise[wv][1] : nullptr;
    const float* nzl = welsh_tp_noise(p, 2) ? s_noise[wv][2] : nullptr;
    // ONE instantiation for retuned and static patches (the retune's tests are scalar branches on the patch flags): a second
    // copy of the unrolled frames is ~3,000 more instructions of kernel text, and this kernel — latency-bound, a handful of
    // wavefronts per CU — pays for text it does not keep in the 64 KB instruction cache two CUs share (round 3, tools/tp_probe.py:
    // the same first 340 instructions took 3,100 cycles in one build and 15,900 in a build 12 % longer).
    welsh_tp_chunk<true, FULL_COEF, CH>(p, s, rc, first0, n0, cnt, nlive, scans, ph1, ph2, nz1, nz2, nzl, cur0, co, coef_full, mine);
  }
  TP_PROBE // 5: pass 2 (feed-forward + affine push)
  // inclusive scan of the affine maps over the voice's lanes, then every lane's start state
  Lp24Affine incl = mine;
  tp_scan_affine<LPV>(incl);
  const double s_init[4] = {s0.filt.s0, s0.filt.s1, s0.filt.s2, s0.filt.s3};
  double s_end[4];
  lp24_affine_mul(incl, s_init, s_end, true);
  double st[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { st[i] = tp_shfl(s_end[i], (int)lane - 1); if (vl == 0) st[i] = s_init[i]; }
  TP_PROBE // 6: affine scan + start state
  float oL[CH], oR[CH];
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    float y = 0.0f;
    if ((co.live >> j) & 1u) {
      const Lp24CoefD cj = retunes ? welsh_tp_coef_at<true, FULL_COEF, CH>(p, co, cur0, coef_full, j) : cur0;
      y = (float)lp24_step_v(st, cj, (double)co.x[j]);
    }
    const float m = y * co.amp[j];
    oL[j] = m * p.gl; oR[j] = m * p.gr;
  }
  TP_PROBE // 7: pass B (filter outputs)
  if constexpr (HEAD_BQ) { // this voice's lane of the BiQuad bank, left then right
    const size_t tn = 2 * (size_t)n;
    const uint32_t vb = VPW == 1 ? v : vme;
    const BiquadCoefD cf{a.bq_coef[vb], a.bq_coef[(size_t)n + vb], a.bq_coef[(size_t)2 * n + vb], a.bq_coef[(size_t)3 * n + vb], a.bq_coef[(size_t)4 * n + vb]};
    const float wm = a.bq_wet[vb];
#pragma unroll
    for (uint32_t ch = 0; ch < 2; ++ch) {
      const size_t t = (size_t)ch * n + vb;
      float xin[CH], y[CH];
      double ns[4];
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) xin[j] = j < cnt ? (ch ? oR[j] : oL[j]) : 0.0f;
      const bool holds_end = bq_tp_wave<CH, LPV>(xin, cnt, lane, frames, cf, a.bq_st[t], a.bq_st[tn + t], a.bq_st[2 * tn + t], a.bq_st[3 * tn + t], wm, y, ns);
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) { if (ch) oR[j] = y[j]; else oL[j] = y[j]; }
      if (voice && holds_end) { a.bq_st[t] = ns[0]; a.bq_st[tn + t] = ns[1]; a.bq_st[2 * tn + t] = ns[2]; a.bq_st[3 * tn + t] = ns[3]; }
    }
  }

  TP_PROBE // 8: BiQuad head
  // ---- outputs
  {
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      s_tile[wv][0][n0 + j] = (voice && j < cnt) ? oL[j] : 0.0f;
      s_tile[wv][1][n0 + j] = (voice && j < cnt) ? oR[j] : 0.0f;
    }
    __syncthreads();
    TP_PROBE // 9: tile + barrier
    // block-writing form: the workgroup's adjacent voices of one (channel, frame) leave as 16-byte stores when the row
    // segment is whole and aligned (instead of 4-byte stores from every wavefront)
    const uint32_t vbase = grp * WGV;
    const bool vec = !FUSED && kTpWaves == 4 && (n & 3u) == 0 && vbase + WGV <= n && (a.ch_stride & 3u) == 0;
    for (uint32_t t = threadIdx.x; t < 2 * frames; t += kTpThreads) {
      const uint32_t ch = t / frames, f = t % frames;
      float q[WGV];
      float acc = 0.0f;
#pragma unroll
      for (uint32_t w = 0; w < WGV; ++w) { q[w] = s_tile[w][ch][f]; acc += q[w]; }
      if (a.rows) a.rows[((size_t)g.row * 2 + ch) * frames + f] = acc; // (null: the block goes straight into an effect chain, which replaces its lane sums)
      if (vec) {
#pragma unroll
        for (uint32_t w = 0; w < WGV; w += 4)
          *reinterpret_cast<float4*>(a.out + ch * a.ch_stride + (size_t)f * n + vbase + w) = make_float4(q[w], q[w + 1], q[w + 2], q[w + 3]);
      }
    }
    if (!FUSED && !vec && voice) {
#pragma unroll
      for (uint32_t j = 0; j < CH; ++j) {
        if (j < cnt) {
          a.out[(size_t)(n0 + j) * n + vme] = oL[j];
          a.out[a.ch_stride + (size_t)(n0 + j) * n + vme] = oR[j];
        }
      }
    }
  }

  TP_PROBE // 10: rows / block stores
  // ---- state after the block: the lane that holds the last frame has every running value
  const uint32_t last = frames ? (frames - 1) / CH : 0u;
  OscState e1 = nz_end, e2 = nz_end, el = nz_end;
  e1.x1 = tp_shfl(nz_end.x1, lbase + 0); e1.x2 = tp_shfl(nz_end.x2, lbase + 0);
  e2.x1 = tp_shfl(nz_end.x1, lbase + 1); e2.x2 = tp_shfl(nz_end.x2, lbase + 1);
  el.x1 = tp_shfl(nz_end.x1, lbase + 2); el.x2 = tp_shfl(nz_end.x2, lbase + 2);
  if (voice && vl == last && frames) {
    s.o1.x1 = e1.x1; s.o1.x2 = e1.x2; s.o2.x1 = e2.x1; s.o2.x2 = e2.x2; s.lfo.x1 = el.x1; s.lfo.x2 = el.x2;
    s.filt.s0 = s_end[0]; s.filt.s1 = s_end[1]; s.filt.s2 = s_end[2]; s.filt.s3 = s_end[3];
    soa_store(a.state, n, vme, s);
  }
  TP_PROBE_END // 11: state store
}
template <bool FUSED, bool HEAD_BQ = false, bool FULL_COEF = false, int VPW = 1>
__global__ __launch_bounds__(kTpThreads, VPW == 1 ? GROOVE_TP_WAVES : 2) void welsh_tp_kernel(TpArgs a) {
  __shared__ WelshTpSmem<VPW> sm;
  welsh_tp_body<FUSED, HEAD_BQ, FULL_COEF, VPW>(a, tp_wg_own(), sm);
}
// FmVoice: one wavefront per voice, 64 lanes x 4 frames (same argument block; rows / out as above).
constexpr uint32_t kFmTpMaxVoices = 131072; // the serial kernel's 256-frame walk costs ~0.09 ms whatever the size; above this it has the wavefronts
// VPW voices per wavefront (round 3): an FM voice's per-wavefront costs (parameters, two envelope seeks, the scan, the tile turn)
// outweigh its frames (two sines and a conversion each), and a big FM bank is throughput-bound — config #5's 32,768 FM voices
// were 32,768 wavefronts beside the Welsh bank that bounds the step.  With four voices per wavefront (16 lanes x 16 frames each;
// parameters per lane: FM patches differ from voice to voice) a voice costs ~40 % fewer issued instructions, and the prefix sum
// of the carrier increments is four DPP row shifts.
template <int VPW> struct FmTpSmem { float tile[kTpWaves * VPW][2][kTpMaxFrames]; };
template <bool FUSED, int VPW>
__device__ __forceinline__ void fm_tp_body(const TpArgs& a, const TpWg g, FmTpSmem<VPW>& sm) {
  static_assert(VPW == 1 || VPW == 2 || VPW == 4, "");
  constexpr uint32_t LPV = 64 / VPW, CH = kTpChunk * VPW, WGV = kTpWaves * VPW;
  float (&s_tile)[WGV][2][kTpMaxFrames] = sm.tile;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t vl = lane & (LPV - 1u), sub = lane / LPV, wv = wave * VPW + sub;
  const uint32_t v0 = g.wg * WGV + wv;
  const bool voice = v0 < a.n;
  const uint32_t vq = voice ? v0 : a.n - 1; // (per lane when VPW > 1)
  const uint32_t v = VPW == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)vq) : vq;
  const uint32_t frames = a.frames, n = a.n;
  const FmParams pl = soa_load<FmParams>(a.params, n, v);
  const FmParams p = VPW == 1 ? make_scalar(pl) : pl; // one voice per wavefront: the patch in SGPRs
  const FmState s0 = soa_load<FmState>(a.state, n, v);
  if constexpr (FUSED) tp_reduce_prev(a.prev, threadIdx.x, g.pwg, g.pn_wg);
  const bool first0 = (s0.vflags & VF_FIRST) != 0;
  const uint32_t live_total = env_idle_at(s0.cenv, p.cenv, frames);
  const uint32_t n0 = vl * CH;
  const uint32_t cnt = n0 < frames ? (frames - n0 < CH ? frames - n0 : CH) : 0u;
  FmState s = s0;
  env_seek(s.cenv, p.cenv, n0 < frames ? n0 : 0u);
  env_seek(s.menv, p.menv, n0 < frames ? n0 : 0u);
  const uint32_t live_before = n0 < live_total ? n0 : live_total;
  const uint64_t adv = (uint64_t)(live_before - ((first0 && live_before >= 1u) ? 1u : 0u));
  uint64_t mph = s0.modulator.phase + adv * s0.m_inc;
  if (live_before >= 1u) s.vflags = 0;
  uint64_t loc[CH], run = 0;
  float cval[CH];
  uint32_t lives = 0;
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    const uint32_t f = n0 + j;
    cval[j] = 0.0f;
    uint64_t inc = 0;
    if (j < cnt) {
      env_tick(s.cenv, p.cenv);
      env_tick(s.menv, p.menv);
      if (f < live_total) {
        lives |= 1u << j;
        const bool is_first = first0 && f == 0;
        if (!is_first) mph += s0.m_inc;
        const uint64_t ci = fm_tp_carrier_inc(p, s0.c_inc, mph, s.menv.value);
        if (!is_first) inc = ci;
        cval[j] = s.cenv.value;
        s.vflags = 0;
      }
    }
    run += inc;
    loc[j] = run;
  }
  uint64_t t = run; // inclusive scan of the lane totals over the voice's lanes
  if constexpr (VPW == 4) { // 16 lanes = one DPP row
    t += tp_dpp_u64<0x111, 0xF>(0, t); t += tp_dpp_u64<0x112, 0xF>(0, t); t += tp_dpp_u64<0x114, 0xF>(0, t); t += tp_dpp_u64<0x118, 0xF>(0, t);
  } else {
    uint64_t unused = 0;
    tp_scan_add_u64<LPV>(t, unused);
  }
  const uint64_t base = s0.carrier.phase + (t - run);
  float oL[CH], oR[CH];
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    float m = 0.0f;
    if ((lives >> j) & 1u) m = osc_value(GROOVE_WAVE_SINE, base + loc[j], 0, 0.0f) * cval[j];
    oL[j] = m * p.gl; oR[j] = m * p.gr;
  }
  {
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      s_tile[wv][0][n0 + j] = (voice && j < cnt) ? oL[j] : 0.0f;
      s_tile[wv][1][n0 + j] = (voice && j < cnt) ? oR[j] : 0.0f;
    }
    __syncthreads();
    for (uint32_t tt = threadIdx.x; tt < 2 * frames; tt += kTpThreads) {
      const uint32_t ch = tt / frames, f = tt % frames;
      float acc = 0.0f;
#pragma unroll
      for (uint32_t w = 0; w < WGV; ++w) acc += s_tile[w][ch][f];
      a.rows[((size_t)g.row * 2 + ch) * frames + f] = acc;
    }
  }
  if (!FUSED && voice) {
#pragma unroll
    for (uint32_t j = 0; j < CH; ++j) {
      if (j < cnt) {
        a.out[(size_t)(n0 + j) * n + v] = oL[j];
        a.out[a.ch_stride + (size_t)(n0 + j) * n + v] = oR[j];
      }
    }
  }
  const uint32_t last = frames ? (frames - 1) / CH : 0u;
  if (voice && vl == last && frames) {
    s.modulator.phase = mph;
    s.carrier.phase = base + run;
    soa_store(a.state, n, v, s);
  }
}
template <bool FUSED, int VPW = 1>
__global__ __launch_bounds__(kTpThreads) void fm_tp_kernel(TpArgs a) {
  __shared__ FmTpSmem<VPW> sm;
  fm_tp_body<FUSED, VPW>(a, tp_wg_own(), sm);
}
// SamplerVoice: pointer stepping is a closed form in the frame index (idx0 + k * step, Q20.44), so a voice's block is a
// pure gather (the serial form walks 256 frames in chunks of 16 fetches: 45 us for a one-wave-per-SIMD bank).  Exact,
// like the serial form.  A wavefront takes `vpw` ADJACENT voices (<= 64), one after the other, its 64 lanes x 4 frames
// each time: the voices' parameters and state are read and written lane-parallel (lane l = the wave's l-th voice:
// coalesced rows, one round trip), handed to the loop by v_readlane, and the voices' fetches are independent, so
// several voices' gathers are in flight together.  Sixteen waves per workgroup and vpw = min(ceil(n / 1024), 16): a bank
// of up to 16,384 voices leaves at most 64 partial rows, which the bus reduction sums in ONE launch (config #4: render +
// two reduction launches -> render + one); bigger banks keep 16 voices per wave and more workgroups.
constexpr uint32_t kSamplerTpMaxVoices = 65536;
inline uint32_t sampler_tp_vpw(uint32_t n) { return n <= 1024 ? 1u : std::min<uint32_t>((n + 1023) / 1024, 16u); } // (32 voices per wave for 32,768 voices: 38 us against ~20 with 16)
inline uint32_t sampler_tp_workgroups(uint32_t n, uint32_t vpw = 0) { const uint32_t per = kSamplerTpWaves * (vpw ? vpw : sampler_tp_vpw(n)); return (n + per - 1) / per; }
// groove_bank_render_mix_deferred: no reduction launch to keep short, so the voices are spread over the chip — at most 512 rows
// (config #4: 4 voices per wavefront, 256 workgroups on 256 CUs instead of 64 on 64)
inline uint32_t sampler_tp_vpw_deferred(uint32_t n) { return n <= 1024 ? 1u : std::min<uint32_t>((n + 8191) / 8192 * 2, 16u); }
template <int WAVES> struct SamplerTpSmem {
  float tile[WAVES][kTpMaxFrames];
  volatile uint32_t ev[WAVES][64]; // volatile: lanes read what OTHER lanes of the wave wrote, with no barrier the compiler knows of
};
// WAVES wavefronts per workgroup: 16 in the bank's own kernel, 4 in the mixed kernel (whose workgroups are 256 threads)
template <bool FUSED, int WAVES>
__device__ __forceinline__ void sampler_tp_body(const TpArgs& a, const float* __restrict__ bank, const InlineEvents& ie, const uint32_t vpw, const TpWg g, SamplerTpSmem<WAVES>& sm) {
  float (&s_tile)[WAVES][kTpMaxFrames] = sm.tile;
  volatile uint32_t (&s_ev)[WAVES][64] = sm.ev;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t frames = a.frames, n = a.n;
  const uint32_t v_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)((g.wg * WAVES + wave) * vpw));
  const uint32_t cnt = v_begin < n ? min(vpw, n - v_begin) : 0u; // this wave's voices (wave-uniform)
  // lane l < cnt: voice v_begin + l
  const bool mine = lane < cnt;
  const uint32_t v = mine ? v_begin + lane : (n - 1);
  const SamplerParams p = soa_load<SamplerParams>(a.params, n, v);
  SamplerState s0 = soa_load<SamplerState>(a.state, n, v);
  if constexpr (FUSED) tp_reduce_prev(a.prev, threadIdx.x, g.pwg, g.pn_wg);
  if (ie.n && cnt) { // this block's note events (strictly increasing voices): those of this wave's voice range
    uint32_t lo = 0, hi = ie.n;
    while (lo < hi) { // first event at or after v_begin (scalar loads)
      const uint32_t mid = (lo + hi) >> 1;
      if (ie.ev[mid].voice < v_begin) lo = mid + 1; else hi = mid;
    }
    s_ev[wave][lane] = 0u;
    if (lo + lane < ie.n) { // at most cnt <= 64 of them fall into the range: one event per lane, scattered to its voice's lane
      const groove_note_event e = ie.ev[lo + lane];
      if (e.voice - v_begin < cnt) s_ev[wave][e.voice - v_begin] = 0x10000u | ((uint32_t)(e.on != 0) << 8) | e.key;
    }
    __builtin_amdgcn_wave_barrier(); // (scheduling fence; a wave's LDS operations complete in issue order)
    const uint32_t got = s_ev[wave][lane];
    if (mine && (got & 0x10000u)) sampler_note(p, s0, got & 0xFFu, (got >> 8) & 1u);
  }
  // frames of the block this lane's voice plays: a prefix, the index only moves forward
  uint32_t valid = 0;
  if (mine && s0.playing && frames) {
    const uint64_t lim = (uint64_t)p.length << 44;
    if (s0.idx < lim) {
      const uint64_t room = lim - s0.idx - 1; // idx + f * step <= lim - 1
      const uint64_t fmax = s0.step ? room / s0.step : (uint64_t)frames;
      valid = fmax >= frames ? frames : (uint32_t)fmax + 1u;
    }
  }
  const uint32_t n0 = lane * kTpChunk;
  float acc[kTpChunk];
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) acc[j] = 0.0f;
  const uint32_t idx_lo = (uint32_t)s0.idx, idx_hi = (uint32_t)(s0.idx >> 32), step_lo = (uint32_t)s0.step, step_hi = (uint32_t)(s0.step >> 32);
  constexpr uint32_t U = 4; // voices whose fetches are issued together
  for (uint32_t k0 = 0; k0 < cnt; k0 += U) { // k is wave-uniform: v_readlane
    uint32_t off[U], len[U], playing[U];
    float gain[U];
    uint64_t idx[U], step[U];
    uint32_t any = 0;
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const int k = (int)min(k0 + u, cnt - 1);
      off[u] = (uint32_t)__builtin_amdgcn_readlane((int)p.offset, k);
      len[u] = (uint32_t)__builtin_amdgcn_readlane((int)p.length, k);
      gain[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p.gain), k));
      idx[u] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)idx_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)idx_lo, k);
      step[u] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)step_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)step_lo, k);
      playing[u] = k0 + u < cnt ? (uint32_t)__builtin_amdgcn_readlane((int)s0.playing, k) : 0u;
      any |= playing[u];
    }
    if (!any) continue; // (uniform)
    float raw[U][kTpChunk];
    bool ok[U][kTpChunk];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
#pragma unroll
      for (uint32_t j = 0; j < kTpChunk; ++j) {
        const uint32_t f = n0 + j;
        const uint32_t i = (uint32_t)((idx[u] + (uint64_t)f * step[u]) >> 44);
        ok[u][j] = playing[u] && f < frames && i < len[u];
        raw[u][j] = bank[(size_t)off[u] + (i < len[u] ? i : len[u] - 1)];
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) { // accumulated voice by voice, in voice order
      if (!playing[u]) continue;
      float x[kTpChunk];
#pragma unroll
      for (uint32_t j = 0; j < kTpChunk; ++j) { x[j] = ok[u][j] ? raw[u][j] * gain[u] : 0.0f; acc[j] += x[j]; }
      if (!FUSED) {
        const uint32_t vk = v_begin + k0 + u;
#pragma unroll
        for (uint32_t j = 0; j < kTpChunk; ++j) {
          if (n0 + j < frames) {
            a.out[(size_t)(n0 + j) * n + vk] = x[j];
            a.out[a.ch_stride + (size_t)(n0 + j) * n + vk] = x[j];
          }
        }
      }
    }
  }
  if (!FUSED) { // voices that do not play still own their columns of the block
    for (uint32_t k = 0; k < cnt; ++k) {
      if (__builtin_amdgcn_readlane((int)s0.playing, (int)k)) continue;
      const uint32_t vk = v_begin + k;
#pragma unroll
      for (uint32_t j = 0; j < kTpChunk; ++j) {
        if (n0 + j < frames) { a.out[(size_t)(n0 + j) * n + vk] = 0.0f; a.out[a.ch_stride + (size_t)(n0 + j) * n + vk] = 0.0f; }
      }
    }
  }
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) s_tile[wave][n0 + j] = acc[j];
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < frames; t += WAVES * 64) {
    float sum = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) sum += s_tile[w][t];
    a.rows[((size_t)g.row * 2 + 0) * frames + t] = sum; // mono voices: the same sum on both channels
    a.rows[((size_t)g.row * 2 + 1) * frames + t] = sum;
  }
  if (mine && frames) {
    SamplerState s = s0;
    s.idx = s0.idx + (uint64_t)valid * s0.step;
    if (s0.playing && valid < frames) s.playing = 0; // ran off the end inside the block
    soa_store(a.state, n, v, s);
  }
}
template <bool FUSED>
__global__ __launch_bounds__(kSamplerTpThreads) void sampler_tp_kernel(TpArgs a, const float* __restrict__ bank, InlineEvents ie, uint32_t vpw) {
  __shared__ SamplerTpSmem<kSamplerTpWaves> sm;
  sampler_tp_body<FUSED, kSamplerTpWaves>(a, bank, ie, vpw, tp_wg_own(), sm);
}
// ------------------------------------------------------------------ several small banks in ONE launch (round 5)
// A project of a few small banks of different kinds (config #5's share of one of eight GPUs: 8,192 Welsh + 4,096 FM + 4,096
// sampler voices) rendered its banks in turn, three launches per block whose durations add (35 + 13 + 7.5 us).  They cannot
// overlap as separate launches either: the Welsh time-parallel kernel's two wavefronts per SIMD hold the whole register file.
// ONE grid of 256-thread workgroups, each dispatching on its index into the Welsh / FM / sampler body (the most expensive kind
// first), ONE row buffer and ONE carried bus reduction (tp_reduce_prev over all of the launch's workgroups) for the project:
// the FM and sampler wavefronts — latency-bound, 7 - 13 us as launches of their own because each is a lone wavefront per SIMD —
// now fill wavefront slots as the Welsh ones retire, and the step costs the sum of the wavefronts' SLOT-time over the chip's
// slots instead of the sum of three launch durations.  (Same register budget for all: the FM / sampler bodies run at the Welsh
// body's occupancy; they are few.)  The bus is one sum over all instruments, orchestrator.rs:397-410.
struct TpMixedBank { const uint32_t* params; uint32_t* state; uint32_t n, vpw, wg0, n_wg; }; // n_wg 0: no bank of this kind
struct TpMixedArgs {
  TpMixedBank welsh, fm, sampler; // the grid: [welsh | fm | sampler] workgroup ranges (wg0 = the range's first workgroup)
  float* rows;                    // rows[workgroup of the grid][ch][frame]
  const float* pcm;               // the sampler bank's sample memory
  RenderConsts rc; uint32_t frames;
  TpPrev prev;
};
#ifndef GROOVE_MIXED_WAVES
#define GROOVE_MIXED_WAVES 2
#endif
template <int WVPW, int FVPW>
__global__ __launch_bounds__(kTpThreads, WVPW == 1 ? GROOVE_TP_WAVES : GROOVE_MIXED_WAVES) void tp_mixed_kernel(TpMixedArgs m, InlineEvents ie) {
  union Smem { WelshTpSmem<WVPW> w; FmTpSmem<FVPW> f; SamplerTpSmem<kTpWaves> s; };
  __shared__ Smem sm;
  const uint32_t i = blockIdx.x;
  TpArgs a{nullptr, nullptr, nullptr, m.rows, 0, m.rc, 0, m.frames};
  a.prev = m.prev;
  if (i - m.welsh.wg0 < m.welsh.n_wg) {
    a.params = m.welsh.params; a.state = m.welsh.state; a.n = m.welsh.n; a.vpw = WVPW;
    welsh_tp_body<true, false, false, WVPW>(a, TpWg{i - m.welsh.wg0, m.welsh.n_wg, i, i, gridDim.x}, sm.w);
  } else if (i - m.fm.wg0 < m.fm.n_wg) {
    a.params = m.fm.params; a.state = m.fm.state; a.n = m.fm.n; a.vpw = FVPW;
    fm_tp_body<true, FVPW>(a, TpWg{i - m.fm.wg0, m.fm.n_wg, i, i, gridDim.x}, sm.f);
  } else {
    a.params = m.sampler.params; a.state = m.sampler.state; a.n = m.sampler.n; a.vpw = m.sampler.vpw;
    sampler_tp_body<true, kTpWaves>(a, m.pcm, ie, m.sampler.vpw, TpWg{i - m.sampler.wg0, m.sampler.n_wg, i, i, gridDim.x}, sm.s);
  }
}
constexpr uint32_t kMixedSamplerVpw = 8; // sampler voices per wavefront in the mixed kernel
inline uint32_t mixed_sampler_workgroups(uint32_t n, uint32_t vpw) { const uint32_t per = kTpWaves * vpw; return (n + per - 1) / per; }
void launch_tp_mixed(const TpMixedArgs& m, const InlineEvents& ie, uint32_t grid, hipStream_t st, hipEvent_t done = nullptr); // welsh.vpw 1 | 2, fm.vpw 1 | 4
void launch_welsh_tp(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr); // a.bq_coef set (block-writing form): the BiQuad head fused; done: an event bound to the dispatch
void launch_fm_tp(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr); // a.vpw voices per wavefront (1, 2, 4)
void launch_sampler_tp(const TpArgs& a, const float* bank, const InlineEvents& ie, hipStream_t st, bool fused, hipEvent_t done = nullptr);
inline uint32_t welsh_tp_workgroups(uint32_t n, uint32_t vpw = 1) { return (n + kTpWaves * vpw - 1) / (kTpWaves * vpw); } // FM: groups of 4 vpw voices per workgroup, plain order
inline uint32_t welsh_tp_grid(uint32_t n, uint32_t vpw = 1) { // Welsh: padded for the XCD-aware mapping (idle workgroups write zero rows)
  const uint32_t g = welsh_tp_workgroups(n, vpw);
  return g >= 16 ? (g + 7u) & ~7u : g;
}
#endif // __HIPCC__

} // namespace groove
