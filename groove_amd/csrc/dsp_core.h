// dsp_core.h — per-lane DSP arithmetic of the MI355X render path.
//
// One voice (or one effect channel) lives in one lane of a 64-wide wavefront; the
// functions here are the per-frame bodies the HIP kernels in kernels.hip loop over
// with all state in registers.  They are plain inline functions so the same text can
// also be compiled by g++ into the host-side numerics harness under tests/emul/ (a
// DEVELOPMENT CHECK of the fp32/f64 arithmetic choices against the f64 oracle; it is
// not a product path and libgroove_hip.so never contains it).
//
// Arithmetic policy:
//   * phases are 64-bit fixed-point turn counters (exact wrap, carry = hard-sync flag);
//   * feed-forward math (waveforms, envelopes, coefficient formulas, pan) is fp32;
//   * IIR recurrences (24 dB low-pass sections, biquads, recirculating combs) are f64:
//     CDNA4 runs v_fma_f64 at half the fp32 rate, which is cheaper than compensating
//     an fp32 recurrence whose poles sit 1e-5 from the unit circle;
//   * anything that moves a waveform EDGE (pitch / pulse-width LFO routing, FM index)
//     is f64, because an edge landing on a different frame is a full-scale error.
//
// Reference items implemented: a1 Oscillator, a2 Envelope, a3/a4
// BiQuad + 24 dB low-pass, a5 WelshVoice, a6 FmVoice, a7 SamplerVoice, a8-a12 effects,
// a13 Dca.  Citations: see include/groove_types.h and oracle/oracle_dsp.hpp.
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/groove_types.h"

#if defined(__HIPCC__)
#define GROOVE_HD __host__ __device__ __forceinline__
#else
#define GROOVE_HD inline
#endif

namespace groove {

// ------------------------------------------------------------------ small math
GROOVE_HD float fast_exp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x); // v_exp_f32
#else
  return exp2f(x);
#endif
}
GROOVE_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rcp_f32: 1 ulp.  (A Newton step on top changed neither the worst nor the median per-voice error
  // against the oracle — tools/gpu_error.py: 2.571e-6 / 1.1e-7 either way — and cost 1.5 % of a block.)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// sin(2*pi*x) for x in [-0.25, 0.25] (turns); odd polynomial in y = 2*pi*x.
GROOVE_HD float sin_turns_folded(float x) {
  const float y = x * 6.283185307179586f;
  const float y2 = y * y;
  float p = 2.5992782e-06f;            // least-squares fit on Chebyshev nodes, |y| <= pi/2:
  p = fmaf(p, y2, -1.9806202e-04f);    // max abs error 1.0e-7 in fp32 (tests/test_emul.py)
  p = fmaf(p, y2, 8.3330103e-03f);
  p = fmaf(p, y2, -1.6666657e-01f);
  return fmaf(p * y2, y, y);
}
GROOVE_HD double sin_turns_folded_f64(double x) {
  const double y = x * 6.28318530717958647692;
  const double y2 = y * y;
  double p = -8.22063524662432971696e-18; // -1/19!
  p = fma(p, y2, 2.81145725434552076320e-15);
  p = fma(p, y2, -7.64716373181981647590e-13);
  p = fma(p, y2, 1.60590438368216145994e-10);
  p = fma(p, y2, -2.50521083854417187751e-08);
  p = fma(p, y2, 2.75573192239858906526e-06);
  p = fma(p, y2, -1.98412698412698412698e-04);
  p = fma(p, y2, 8.33333333333333333333e-03);
  p = fma(p, y2, -1.66666666666666666667e-01);
  return fma(p * y2, y, y);
}
// Fold a signed 32-bit phase (turns * 2^32, [-0.5, 0.5)) into [-0.25, 0.25] so that
// sin(2 pi p) is unchanged: p -> 0.5 - p (or -0.5 - p) when |p| > 0.25.
GROOVE_HD int32_t fold_quarter(int32_t q) {
  // top two bits differ <=> |p| >= 0.25
  if ((q ^ (q << 1)) < 0) q = (int32_t)(0x80000000u - (uint32_t)q);
  return q;
}
GROOVE_HD int64_t fold_quarter64(int64_t q) {
  if ((q ^ (q << 1)) < 0) q = (int64_t)(0x8000000000000000ull - (uint64_t)q);
  return q;
}
// 2^x for |x| <= 1 in f64 (Taylor in x ln2, 1e-14 relative).
GROOVE_HD double exp2_small_f64(double x) {
  const double t = x * 0.693147180559945309417;
  double p = 1.14707455977297247139e-11; // 1/14!
  p = fma(p, t, 1.60590438368216145994e-10);
  p = fma(p, t, 2.08767569878680989792e-09);
  p = fma(p, t, 2.50521083854417187751e-08);
  p = fma(p, t, 2.75573192239858906526e-07);
  p = fma(p, t, 2.75573192239858906526e-06);
  p = fma(p, t, 2.48015873015873015873e-05);
  p = fma(p, t, 1.98412698412698412698e-04);
  p = fma(p, t, 1.38888888888888888889e-03);
  p = fma(p, t, 8.33333333333333333333e-03);
  p = fma(p, t, 4.16666666666666666667e-02);
  p = fma(p, t, 1.66666666666666666667e-01);
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  return fma(p, t, 1.0);
}
// e^x for |x| <= 1.5e-3 (x^5/120 < 1e-16): the per-frame growth factor of a slowly moving exponent.
GROOVE_HD double exp_tiny_f64(double x) {
  double p = 4.16666666666666666667e-02;
  p = fma(p, x, 1.66666666666666666667e-01);
  p = fma(p, x, 0.5);
  p = fma(p, x, 1.0);
  return fma(p, x, 1.0);
}
// tan of the angle x in (0, pi/2) REDUCED to [0, pi/4]: returns t = tan(min(x, pi/2 - x)) <= 1 and
// hi = (x > pi/4), i.e. tan(x) = hi ? 1/t : t.  The caller keeps working with t (never forms
// 1/t), which is what keeps the filter coefficients accurate next to Nyquist.
// tan(z) = z P(z^2) on [0, pi/4]: weighted least squares on Chebyshev nodes for the RELATIVE error
// (the coefficient formulas need t to a relative accuracy); max relative error 1.5e-7 in fp32, the
// same as the sin / cos quotient it replaces, in 8 instructions instead of 14 and no reciprocal.
GROOVE_HD float tan_of_reduced(float z);
GROOVE_HD float tan_reduced(float x, bool& hi) {
  hi = x > 0.78539816339744831f;
  return tan_of_reduced(hi ? (1.57079632679489662f - x) : x);
}
GROOVE_HD float tan_of_reduced(float z) {
  const float z2 = z * z;
  float p = 9.449327447e-03f;
  p = fmaf(p, z2, 2.985451510e-03f);
  p = fmaf(p, z2, 2.453938616e-02f);
  p = fmaf(p, z2, 5.336849955e-02f);
  p = fmaf(p, z2, 1.333961619e-01f);
  p = fmaf(p, z2, 3.333309016e-01f);
  p = fmaf(p, z2, 1.000000015e+00f);
  return p * z;
}
GROOVE_HD float clamp01f(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
GROOVE_HD double clamp01d(double x) { return fmin(fmax(x, 0.0), 1.0); }

// f64 -> u64 (truncating) for 0 <= x < 2^64
GROOVE_HD uint64_t f64_to_u64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // two v_cvt_u32_f64 (truncating) around one exact fma: hi = floor(x / 2^32), lo = x - hi * 2^32
  // (exact: both are multiples of ulp(x) below 2^32) -- 5 instructions instead of the generic 7.
  const uint32_t hi = (uint32_t)(x * 2.3283064365386963e-10);
  const uint32_t lo = (uint32_t)fma((double)hi, -4294967296.0, x);
  return ((uint64_t)hi << 32) | (uint64_t)lo;
#else
  return (uint64_t)x;
#endif
}
// turns (f64, any sign, |x| < 2^31) -> wrapped 64-bit phase increment
GROOVE_HD uint64_t turns_to_phase(double turns) {
  double fl = floor(turns);
  double frac = turns - fl; // [0,1)
  return (uint64_t)(frac * 18446744073709551616.0);
}
// signed increment: turns may be negative (FM through zero); two's complement add.  Whole turns drop out (the oracle's pos -= floor(pos)):
// an FM index of 100 on A4 — the reference's own demo project — swings the carrier past a whole turn per frame, and a u64 conversion of
// 2^64 or more is not a wrap (docs/HISTORY.md section 10 item 29).  |turns| < 1 passes through fract() unchanged: the same bits as before.
GROOVE_HD double fract_f64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fract(x);
#else
  return x - floor(x);
#endif
}
GROOVE_HD uint64_t turns_to_inc(double turns) {
  if (turns >= 0.0) return (uint64_t)(fract_f64(turns) * 18446744073709551616.0);
  return (uint64_t)0 - (uint64_t)(fract_f64(-turns) * 18446744073709551616.0);
}

// ------------------------------------------------------------------ Oscillator (a1)
struct OscState {
  uint64_t phase;  // turns * 2^64
  uint32_t x1, x2; // noise generator
};
GROOVE_HD void osc_reset(OscState& s) { s.phase = 0; s.x1 = 0x70f4f854u; s.x2 = 0xe1e9f0a7u; }
// Voice-level flag word (all oscillators of a voice tick together, so "first tick after
// reset emits position 0" is one bit per voice; note events land at block starts, so the
// bit can only be set on frame 0 of a render call — the kernels peel that frame).
enum : uint32_t { VF_FIRST = 1u };

// Waveform value at `phase` (fp32).  duty64 = duty * 2^64.
GROOVE_HD float osc_value(uint32_t waveform, uint64_t phase, uint64_t duty64, float noise_value) {
  const int32_t q = (int32_t)(uint32_t)(phase >> 32); // signed turns * 2^32
  switch (waveform) {
    case GROOVE_WAVE_SINE:
      return sin_turns_folded((float)fold_quarter(q) * 2.3283064365386963e-10f);
    case GROOVE_WAVE_SQUARE:
    case GROOVE_WAVE_PULSE_WIDTH: return phase < duty64 ? 1.0f : -1.0f;
    case GROOVE_WAVE_TRIANGLE: return fmaf(fabsf((float)q * 2.3283064365386963e-10f), 4.0f, -1.0f);
    case GROOVE_WAVE_SAWTOOTH: return (float)q * 4.6566128730773926e-10f;
    case GROOVE_WAVE_TRIANGLE_SINE: {
      const int32_t r = (int32_t)((uint32_t)q + 0x40000000u);
      return fmaf(fabsf((float)r * 2.3283064365386963e-10f), 4.0f, -1.0f);
    }
    case GROOVE_WAVE_NOISE: return noise_value;
    case GROOVE_WAVE_DEBUG_MAX: return 1.0f;
    case GROOVE_WAVE_DEBUG_MIN: return -1.0f;
    default: return 0.0f;
  }
}
// Same in f64 from the full 64-bit phase (LFO on edge-moving routings).
GROOVE_HD double osc_value_f64(uint32_t waveform, uint64_t phase, uint64_t duty64, float noise_value) {
  const int64_t q = (int64_t)phase;
  const double k = 5.42101086242752217004e-20; // 2^-64
  switch (waveform) {
    case GROOVE_WAVE_SINE: return sin_turns_folded_f64((double)fold_quarter64(q) * k);
    case GROOVE_WAVE_SQUARE:
    case GROOVE_WAVE_PULSE_WIDTH: return phase < duty64 ? 1.0 : -1.0;
    case GROOVE_WAVE_TRIANGLE: return fma(fabs((double)q * k), 4.0, -1.0);
    case GROOVE_WAVE_SAWTOOTH: return (double)q * (2.0 * k);
    case GROOVE_WAVE_TRIANGLE_SINE: {
      const int64_t r = (int64_t)((uint64_t)q + 0x4000000000000000ull);
      return fma(fabs((double)r * k), 4.0, -1.0);
    }
    case GROOVE_WAVE_NOISE: return (double)noise_value;
    case GROOVE_WAVE_DEBUG_MAX: return 1.0;
    case GROOVE_WAVE_DEBUG_MIN: return -1.0;
    default: return 0.0;
  }
}
// musicdsp "fast white noise": integer generator, bit-exact on every platform.
GROOVE_HD float noise_tick(OscState& s) {
  s.x1 ^= s.x2;
  const float v = (float)(int32_t)s.x2 * 4.6566128730773926e-10f; // 2^-31
  s.x2 += s.x1;
  return v;
}

// ------------------------------------------------------------------ Envelope (a2)
enum : uint32_t { ENV_IDLE = 0, ENV_ATTACK = 1, ENV_DECAY = 2, ENV_SUSTAIN = 3, ENV_RELEASE = 4 };
struct EnvParams {
  float attack_len;   // attack_s * SR           (frames for a 0 -> 1 rise)
  uint32_t attack_N;  // ceil(attack_len), computed in f64 on the host
  float decay_len;    // decay_s * SR * (1 - sustain)
  uint32_t decay_N;
  float sustain;
  float release_len;  // release_s * SR          (frames for a 1 -> 0 fall)
};
// value = A + D * (2t - t^2), t = n * inv_len.  Plateaus (IDLE, SUSTAIN) are ramps with
// D = 0 and N = UINT32_MAX, so the per-frame body is branch-free except the stage boundary.
struct EnvState {
  uint32_t state, n, N;
  float A, D, inv_len, value;
};
// N = ceil(len (1 - 2^-16)) (docs/DSP_SPEC.md section 3): a hair below `len`, so that a length that is a whole number of frames in exact
// arithmetic keeps that count on either side of its binary rounding — fp32 here, f64 in the oracle.  With a plain ceil a voice
// released from a round sustain level (0.3 s x 44,100 x 0.6 = 7,938 frames; 7,938.0003 in fp32) went idle one frame later here than in
// the oracle, froze its oscillators a frame apart and came back from its next note-on 0.3 of full scale off (round 5).
GROOVE_HD uint32_t env_frames(float len) {
  if (!(len > 0.0f)) return 0u;
  const float c = ceilf(len * (1.0f - 1.0f / 65536.0f));
  return c > 4.0e9f ? 4000000000u : (uint32_t)c;
}
GROOVE_HD void env_enter_len(EnvState& s, uint32_t st, float from, float to, float len, uint32_t N) {
  s.state = st; s.n = 0; s.A = from; s.D = to - from; s.N = N;
  s.inv_len = len > 0.0f ? 1.0f / len : 0.0f;
}
GROOVE_HD void env_plateau(EnvState& s, uint32_t st, float level) {
  s.state = st; s.n = 0; s.N = 0xFFFFFFFFu; s.A = level; s.D = 0.0f; s.inv_len = 0.0f;
}
GROOVE_HD void env_init(EnvState& s) { env_plateau(s, ENV_IDLE, 0.0f); s.value = 0.0f; }
GROOVE_HD void env_trigger_attack(EnvState& s, const EnvParams& p) {
  const float from = s.value;
  if (from == 0.0f) env_enter_len(s, ENV_ATTACK, 0.0f, 1.0f, p.attack_len, p.attack_N);
  else { const float len = p.attack_len * (1.0f - from); env_enter_len(s, ENV_ATTACK, from, 1.0f, len, env_frames(len)); }
}
GROOVE_HD void env_trigger_release(EnvState& s, const EnvParams& p) {
  if (s.state == ENV_IDLE) return;
  const float from = s.value;
  const float len = p.release_len * from;
  env_enter_len(s, ENV_RELEASE, from, 0.0f, len, env_frames(len));
}
// A tick = the stage-boundary check, then the value at the stage's frame counter.  The two halves are
// also used apart: between two boundaries (of any lane of a wave) a run of frames needs no checks.
GROOVE_HD void env_boundary(EnvState& s, const EnvParams& p) {
  if (s.n >= s.N) { // stage boundary (rare): up to two chained transitions in one frame
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (s.n >= s.N) {
        if (s.state == ENV_ATTACK) env_enter_len(s, ENV_DECAY, 1.0f, p.sustain, p.decay_len, p.decay_N);
        else if (s.state == ENV_DECAY) env_plateau(s, ENV_SUSTAIN, p.sustain);
        else if (s.state == ENV_RELEASE) env_plateau(s, ENV_IDLE, 0.0f);
        else s.n = 0; // plateau counter wrapped after 2^32 frames
      }
    }
  }
}
// The stage's shape A + D (2t - t^2), t = n / len, as a polynomial in the frame counter: n (c1 - c2 n) + A with c1 = 2 D / len,
// c2 = D / len^2.  Every form evaluates THESE operations in THIS order (the hoisted frames of the uniform kernels keep c1 and c2 for
// a segment — welsh_segment_start_hoisted — and pay two operations per envelope and frame instead of four), so the forms stay
// bit-identical; no accumulation, within an ulp or two of the factored form (round 5; the oracle is f64 and does not care).
GROOVE_HD void env_shape_consts(float D, float inv_len, float& c1, float& c2) {
  const float di = D * inv_len;
  c1 = di + di; c2 = di * inv_len;
}
GROOVE_HD float env_shape(float n, float A, float c1, float c2) { return fmaf(n, fmaf(-c2, n, c1), A); }
GROOVE_HD void env_advance(EnvState& s) {
  float c1, c2;
  env_shape_consts(s.D, s.inv_len, c1, c2);
  s.value = env_shape((float)s.n, s.A, c1, c2);
  s.n += 1;
}
// value of the most recent tick (the stage counter has already moved past it)
GROOVE_HD float env_last_value_of(const EnvState& s) {
  float c1, c2;
  env_shape_consts(s.D, s.inv_len, c1, c2);
  return env_shape((float)(s.n - 1u), s.A, c1, c2);
}
GROOVE_HD void env_tick(EnvState& s, const EnvParams& p) {
  env_boundary(s, p);
  env_advance(s);
}
// frames this envelope can advance before its next boundary check must run (>= 1 right after env_boundary)
GROOVE_HD uint32_t env_frames_to_boundary(const EnvState& s) { return s.N - s.n; }

// Launch-wide constants of the render kernels (sample-rate dependent).
struct RenderConsts {
  float pi_over_sr; // pi / SR
  float fc_max;     // 0.49 * SR
  // per-frame retune: x = pi fc / SR of fc = 25 * 800^pct is ONE exp2 with the constants folded into its argument
  float log2_x0;    // log2(25 pi / SR)
  float x_lo, x_hi; // pi / SR (fc = 1 Hz), 0.49 pi (fc = 0.49 SR): the clamp of fc, applied to x
  // two addends of the retune's tangent polynomial: carried here so that a kernel can pin them in registers for the
  // whole block (as literals the compiler re-materialises them with a v_mov on every frame)
  float tan_k1 = 1.333961619e-01f, tan_k2 = 2.453938616e-02f;
  // which look-aheads the uniform kernels may use (kernels.h): bit 0 the filter coefficients', bit 1 the LFO's.  (A run-time word
  // so that the tests can render the same bank with and without: groove_set_look_ahead.)
  uint32_t look = 3u;
};
GROOVE_HD RenderConsts render_consts(double sr) {
  RenderConsts rc;
  rc.pi_over_sr = (float)(3.14159265358979323846 / sr);
  rc.fc_max = (float)(0.49 * sr);
  rc.log2_x0 = (float)log2(25.0 * 3.14159265358979323846 / sr);
  rc.x_lo = (float)(3.14159265358979323846 / sr);
  rc.x_hi = (float)(0.49 * 3.14159265358979323846);
  return rc;
}
// ------------------------------------------------------------------ 24 dB low-pass (a4)
// Per-voice constants: c0 = 1/(cosh^2 r - 0.8535..), d1 = c0 sinh r 1.8477..,
// c2 = 1/(cosh^2 r - 0.1464..), d3 = c2 sinh r 0.7653..  (host, f64 -> f32).
struct Lp24Consts { float c0, d1, c2, d3; };
// Coefficients in "small quantity" form.  With k = tan(pi fc / SR) each section is
//   a1 = 2 (c - k^2) / D,  a2 = (d k - k^2 - c) / D,  b0 = k^2 / D,  D = c + d k + k^2.
// Below SR/4 (k <= 1, poles towards z = +1) fp32 keeps the distance from the unit circle if
// we carry q1 = 2 - a1, q2 = 1 + a2; above SR/4 (k > 1, poles towards z = -1) the same is true
// of q1 = 2 + a1, q2 = 1 + a2 written in t = 1/k.  Both cases are one formula in
// t = min(k, 1/k):   D' = Q + d t + P,  b0 = N / D',  q1 = 2 (d t + 2 P) / D',  q2 = 2 d t / D'
// with (P, Q, N) = (t^2, c, t^2) below and (c t^2, 1, 1) above; a1 = sgn (2 - q1), a2 = q2 - 1.
struct Lp24StateD { double s0, s1, s2, s3; };
// f64 coefficient set derived from the fp32 small-quantity form (exact conversions).
struct Lp24CoefD { double b0a, a1a, a2a, b0b, a1b, a2b; };
// Coefficients for cutoff fc (Hz); pi_over_sr = pi / SR; fc clamped to [1, 0.49 SR].  The fp32 quotients, widened exactly; the
// upper side of SR/4 is an exec-mask region that a wave with no lane above SR/4 skips.
// t and its side of SR/4 for cutoff fc (Hz).  AT the clamp (fc >= 0.49 SR) the reduced angle is 0.01 pi, given as that: the product
// fc_max x (pi / SR) carries two fp32 roundings of an x next to pi/2, which leave pi/2 - x only 3e-6 of relative precision — and a
// high-ripple section thrown to the clamp by a square LFO played 2e-5 of its level off through that (docs/HISTORY.md section 10 item 31).
GROOVE_HD float lp24_t_from_fc(float fc, float pi_over_sr, float fc_max, bool& hi) {
  fc = fminf(fmaxf(fc, 1.0f), fc_max);
  const float t = tan_reduced(fc * pi_over_sr, hi);
  if (fc >= fc_max) { hi = true; return tan_of_reduced(3.14159265358979324e-02f); }
  return t;
}
GROOVE_HD Lp24CoefD lp24_coefd_from_fc(const Lp24Consts& c, float fc, float pi_over_sr, float fc_max) {
  bool hi;
  const float t = lp24_t_from_fc(fc, pi_over_sr, fc_max, hi);
  const float T2 = t * t;
  const float dta = c.d1 * t, dtb = c.d3 * t;
  Lp24CoefD d;
  if (!hi) {
    // q1 = 2 - a1 is the small quantity while the pole pair sits towards z = +1 (k^2 <= c + d k).  A section whose c is far below
    // k^2 (ripples above ~4: c = 1 / (cosh^2 r - ..) ~ 1e-3 .. 1e-9) has its poles towards z = -1 on THIS side of SR/4 too: b0 -> 1,
    // q1 -> 4, and 2 - q1 cancels (a reference patch with ripple 7.1 under a cutoff sweep played 6e-5 off the oracle through that;
    // docs/HISTORY.md section 10 item 27).  There the small quantity is 2 + a1 = q2 + 4 c / D.
    const float sa = c.c0 + dta, sb = c.c2 + dtb;
    const float ia = fast_rcp(sa + T2);
    const float ib = fast_rcp(sb + T2);
    const double q2a = (double)(2.0f * dta * ia), q2b = (double)(2.0f * dtb * ib);
    d.b0a = (double)(T2 * ia); d.a2a = q2a - 1.0;
    d.b0b = (double)(T2 * ib); d.a2b = q2b - 1.0;
    d.a1a = T2 > sa ? fma(4.0, (double)(c.c0 * ia), q2a - 2.0) : 2.0 - (double)(2.0f * (dta + 2.0f * T2) * ia);
    d.a1b = T2 > sb ? fma(4.0, (double)(c.c2 * ib), q2b - 2.0) : 2.0 - (double)(2.0f * (dtb + 2.0f * T2) * ib);
  } else {
    const float Pa = c.c0 * T2, Pb = c.c2 * T2;
    const float ia = fast_rcp(1.0f + dta + Pa);
    const float ib = fast_rcp(1.0f + dtb + Pb);
    d.b0a = (double)(1.0f * ia); d.a1a = (double)(2.0f * (dta + 2.0f * Pa) * ia) - 2.0; d.a2a = (double)(2.0f * dta * ia) - 1.0;
    d.b0b = (double)(1.0f * ib); d.a1b = (double)(2.0f * (dtb + 2.0f * Pb) * ib) - 2.0; d.a2b = (double)(2.0f * dtb * ib) - 1.0;
  }
  return d;
}
// The per-frame retune of a voice: coefficients for cutoff PERCENT pct (fc = 25 * 800^pct Hz, clamped to
// [1 Hz, 0.49 SR]).  Same mathematics as lp24_coefd_from_fc(c, 25 * 800^pct, ...), arranged for the frame loop:
//   * x = pi fc / SR straight from one exp2 (constants folded into the exponent), the clamp of fc applied to x;
//   * tan's argument is min(x, pi/2 - x);
//   * q1 = q2 + 4 b0 (lower side; q2 + 4 P / D' on the upper side), so a1 = +-(2 - q1) is one f64 fma on the widened
//     b0 and q2 instead of five more fp32 operations and a third conversion per section.
GROOVE_HD float med3f(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
  return fminf(fmaxf(x, lo), hi);
#endif
}
// (In two halves, so that the role-split kernel — welsh_split.h — can run them on different wavefronts: the tangent of the
// cutoff, and the coefficients from it.  lp24_coefd_from_pct is their composition: same operations, same order.)
GROOVE_HD float lp24_t_from_pct(float pct, const RenderConsts& rc, bool& hi) {
  const float x = med3f(fast_exp2(fmaf(clamp01f(pct), 9.6438561897747244f, rc.log2_x0)), rc.x_lo, rc.x_hi);
  hi = x > 0.78539816339744831f;
  const float z = fminf(x, 1.57079632679489662f - x);
  // tan(z) = z P(z^2), the polynomial of tan_reduced by Estrin's scheme (four dependent steps instead of seven)
  const float w = z * z, w2 = w * w;
  const float e0 = fmaf(3.333309016e-01f, w, 1.000000015e+00f), e1 = fmaf(5.336849955e-02f, w, rc.tan_k1);
  const float e2 = fmaf(2.985451510e-03f, w, rc.tan_k2);
  const float w4 = w2 * w2;
  const float f0 = fmaf(e1, w2, e0), f1 = fmaf(9.449327447e-03f, w2, e2);
  return fmaf(f1, w4, f0) * z;
}
// `wide` (wave-uniform; WF_COEF_WIDE patches — a section constant c below 1/512, ripple above ~3.8): the lower side's a1 in its TWO-SIDED
// form.  With c that small the pole pair sits towards z = -1 on the lower side of SR/4 too as soon as k^2 > c + d k: b0 -> 1, and
// 2 - q2 - 4 b0 cancels (the rounding of b0, 2^-24, against a distance from -2 of 1e-3: a reference patch with ripple 7.1 under a cutoff
// sweep played 6e-5 off the oracle through that, docs/HISTORY.md section 10 item 27).  There the small quantity is 2 + a1 = q2 + 4 c / D,
// the same algebra (4 b0 + 4 c / D + 2 q2 = 4) with the rounding on the small term.  The side is chosen on the quotients themselves,
// b0 > c / D + q2 / 2 (== k^2 > c + d k), so that the four-role kernel — which hands the quotients from one wavefront to another —
// makes the same choice from the same bits.  Until round 6 such patches ran in the exact-f64 kind (lp24_coefd_from_fc's form of the same
// idea): 14 of the reference's 106 patch files, 13 % of a library-proportioned bank in the slowest kernel for their filter's sake alone.
GROOVE_HD Lp24CoefD lp24_coefd_from_t(const Lp24Consts& c, float t, bool hi, bool wide = false) {
  const float T2 = t * t;
  const float dta = c.d1 * t, dtb = c.d3 * t;
  Lp24CoefD d;
  if (!hi) {
    const float ia = fast_rcp(c.c0 + dta + T2);
    const float ib = fast_rcp(c.c2 + dtb + T2);
    const float fba = T2 * ia, fqa = (dta + dta) * ia, fbb = T2 * ib, fqb = (dtb + dtb) * ib;
    const double b0a = (double)fba, q2a = (double)fqa;
    const double b0b = (double)fbb, q2b = (double)fqb;
    d.b0a = b0a; d.a1a = fma(-4.0, b0a, 2.0 - q2a); d.a2a = q2a - 1.0;
    d.b0b = b0b; d.a1b = fma(-4.0, b0b, 2.0 - q2b); d.a2b = q2b - 1.0;
    if (wide) {
      const float pa = c.c0 * ia, pb = c.c2 * ib;
      if (fba > fmaf(0.5f, fqa, pa)) d.a1a = fma(4.0, (double)pa, q2a - 2.0);
      if (fbb > fmaf(0.5f, fqb, pb)) d.a1b = fma(4.0, (double)pb, q2b - 2.0);
    }
  } else {
    const float Pa = c.c0 * T2, Pb = c.c2 * T2;
    const float ia = fast_rcp(1.0f + dta + Pa);
    const float ib = fast_rcp(1.0f + dtb + Pb);
    const double q2a = (double)((dta + dta) * ia), pa = (double)(Pa * ia);
    const double q2b = (double)((dtb + dtb) * ib), pb = (double)(Pb * ib);
    d.b0a = (double)ia; d.a1a = fma(4.0, pa, q2a - 2.0); d.a2a = q2a - 1.0;
    d.b0b = (double)ib; d.a1b = fma(4.0, pb, q2b - 2.0); d.a2b = q2b - 1.0;
  }
  return d;
}
// lp24_coefd_from_t in two halves again (the four-role kernel): the fp32 quotients, and their widening.  Same operations on the
// same values: lp24_coefd_from_q(lp24_coefq_from_t(c, t, hi), hi) == lp24_coefd_from_t(c, t, hi) bit for bit.
struct Lp24CoefQ { float ba, qa, bb, qb, pa, pb; }; // per section: b0 (upper side: 1 / D'), q2, and on the upper side P / D'
GROOVE_HD Lp24CoefQ lp24_coefq_from_t(const Lp24Consts& c, float t, bool hi, bool wide = false) {
  const float T2 = t * t;
  const float dta = c.d1 * t, dtb = c.d3 * t;
  Lp24CoefQ q;
  if (!hi) {
    const float ia = fast_rcp(c.c0 + dta + T2);
    const float ib = fast_rcp(c.c2 + dtb + T2);
    q.ba = T2 * ia; q.qa = (dta + dta) * ia; q.pa = 0.0f;
    q.bb = T2 * ib; q.qb = (dtb + dtb) * ib; q.pb = 0.0f;
    if (wide) { q.pa = c.c0 * ia; q.pb = c.c2 * ib; } // (lower side, two-sided form: c / D)
  } else {
    const float Pa = c.c0 * T2, Pb = c.c2 * T2;
    const float ia = fast_rcp(1.0f + dta + Pa);
    const float ib = fast_rcp(1.0f + dtb + Pb);
    q.ba = ia; q.qa = (dta + dta) * ia; q.pa = Pa * ia;
    q.bb = ib; q.qb = (dtb + dtb) * ib; q.pb = Pb * ib;
  }
  return q;
}
GROOVE_HD Lp24CoefD lp24_coefd_from_q(const Lp24CoefQ& q, bool hi, bool wide = false) {
  Lp24CoefD d;
  const double q2a = (double)q.qa, q2b = (double)q.qb;
  if (!hi) {
    const double b0a = (double)q.ba, b0b = (double)q.bb;
    d.b0a = b0a; d.a1a = fma(-4.0, b0a, 2.0 - q2a); d.a2a = q2a - 1.0;
    d.b0b = b0b; d.a1b = fma(-4.0, b0b, 2.0 - q2b); d.a2b = q2b - 1.0;
    if (wide) {
      if (q.ba > fmaf(0.5f, q.qa, q.pa)) d.a1a = fma(4.0, (double)q.pa, q2a - 2.0);
      if (q.bb > fmaf(0.5f, q.qb, q.pb)) d.a1b = fma(4.0, (double)q.pb, q2b - 2.0);
    }
  } else {
    const double pa = (double)q.pa, pb = (double)q.pb;
    d.b0a = (double)q.ba; d.a1a = fma(4.0, pa, q2a - 2.0); d.a2a = q2a - 1.0;
    d.b0b = (double)q.bb; d.a1b = fma(4.0, pb, q2b - 2.0); d.a2b = q2b - 1.0;
  }
  return d;
}
GROOVE_HD Lp24CoefD lp24_coefd_from_pct(const Lp24Consts& c, float pct, const RenderConsts& rc, bool wide = false) {
  bool hi;
  const float t = lp24_t_from_pct(pct, rc, hi);
  return lp24_coefd_from_t(c, t, hi, wide);
}
#ifdef GROOVE_EMUL_F32_FILTER_HOOK
// The same two transposed-direct-form-II sections with fp32 state and fp32 arithmetic (the state fields hold float values).
// form 1: plain coefficients a1, a2 rounded to fp32 (ten operations a frame, what an fp32 kernel kind would issue);
// form 2: the "small quantity" form q1 = 2 - |a1|, q2 = 1 + a2 kept in fp32 (fourteen operations a frame).
extern int groove_emul_f32_filter;
inline double groove_emul_lp24_step_f32(Lp24StateD& s, const Lp24CoefD& c, double xd, int form) {
  const float x = (float)xd;
  float st[4] = {(float)s.s0, (float)s.s1, (float)s.s2, (float)s.s3};
  const double b0[2] = {c.b0a, c.b0b}, a1[2] = {c.a1a, c.a1b}, a2[2] = {c.a2a, c.a2b};
  float in = x;
  for (int k = 0; k < 2; ++k) {
    const float b = (float)b0[k];
    const float bx = b * in;
    const float y = bx + st[2 * k];
    if (form == 1) {
      const float fa1 = (float)a1[k], fa2 = (float)a2[k];
      st[2 * k] = fmaf(fa1, y, fmaf(2.0f, bx, st[2 * k + 1]));
      st[2 * k + 1] = fmaf(fa2, y, bx);
    } else {
      const float sg = a1[k] < 0.0 ? -1.0f : 1.0f;
      const float q1 = (float)(2.0 - (a1[k] < 0.0 ? -a1[k] : a1[k])), q2 = (float)(a2[k] + 1.0);
      const float t = fmaf(2.0f, bx, st[2 * k + 1]);
      st[2 * k] = fmaf(-sg * q1, y, fmaf(2.0f * sg, y, t));
      st[2 * k + 1] = fmaf(q2, y, bx - y);
    }
    in = y;
  }
  s.s0 = st[0]; s.s1 = st[1]; s.s2 = st[2]; s.s3 = st[3];
  return (double)in;
}
#endif
// SCALAR_COEF: the coefficients are wave-uniform values the caller keeps in SGPRs (device only).
template <bool SCALAR_COEF = false>
GROOVE_HD double lp24_step(Lp24StateD& s, const Lp24CoefD& c, double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // The ten operations below, written out: left to itself the compiler picks the two-operand
  // v_fmac_f64 for the state updates and then copies all four loop-carried state pairs (and two
  // temporaries) back with v_mov_b64 every frame — 6 of 16 instructions.  Three-operand v_fma_f64
  // updates each state pair in place.  Same operations, same order, same roundings as the C below.
  double bx, y1, t, by, y2, u;
  if constexpr (SCALAR_COEF) {
    asm("v_mul_f64 %[bx], %[b0a], %[x]\n\t"
        "v_add_f64 %[y1], %[bx], %[s0]\n\t"
        "v_fma_f64 %[t], %[bx], 2.0, %[s1]\n\t"
        "v_fma_f64 %[s0], %[a1a], %[y1], %[t]\n\t"
        "v_fma_f64 %[s1], %[a2a], %[y1], %[bx]\n\t"
        "v_mul_f64 %[by], %[b0b], %[y1]\n\t"
        "v_add_f64 %[y2], %[by], %[s2]\n\t"
        "v_fma_f64 %[u], %[by], 2.0, %[s3]\n\t"
        "v_fma_f64 %[s2], %[a1b], %[y2], %[u]\n\t"
        "v_fma_f64 %[s3], %[a2b], %[y2], %[by]"
        : [s0] "+v"(s.s0), [s1] "+v"(s.s1), [s2] "+v"(s.s2), [s3] "+v"(s.s3), [bx] "=&v"(bx), [y1] "=&v"(y1), [t] "=&v"(t),
          [by] "=&v"(by), [y2] "=&v"(y2), [u] "=&v"(u)
        : [x] "v"(x), [b0a] "s"(c.b0a), [a1a] "s"(c.a1a), [a2a] "s"(c.a2a), [b0b] "s"(c.b0b), [a1b] "s"(c.a1b), [a2b] "s"(c.a2b));
  } else {
    asm("v_mul_f64 %[bx], %[b0a], %[x]\n\t"
        "v_add_f64 %[y1], %[bx], %[s0]\n\t"
        "v_fma_f64 %[t], %[bx], 2.0, %[s1]\n\t"
        "v_fma_f64 %[s0], %[a1a], %[y1], %[t]\n\t"
        "v_fma_f64 %[s1], %[a2a], %[y1], %[bx]\n\t"
        "v_mul_f64 %[by], %[b0b], %[y1]\n\t"
        "v_add_f64 %[y2], %[by], %[s2]\n\t"
        "v_fma_f64 %[u], %[by], 2.0, %[s3]\n\t"
        "v_fma_f64 %[s2], %[a1b], %[y2], %[u]\n\t"
        "v_fma_f64 %[s3], %[a2b], %[y2], %[by]"
        : [s0] "+v"(s.s0), [s1] "+v"(s.s1), [s2] "+v"(s.s2), [s3] "+v"(s.s3), [bx] "=&v"(bx), [y1] "=&v"(y1), [t] "=&v"(t),
          [by] "=&v"(by), [y2] "=&v"(y2), [u] "=&v"(u)
        : [x] "v"(x), [b0a] "v"(c.b0a), [a1a] "v"(c.a1a), [a2a] "v"(c.a2a), [b0b] "v"(c.b0b), [a1b] "v"(c.a1b), [a2b] "v"(c.a2b));
  }
  return y2;
#else
#ifdef GROOVE_EMUL_F32_FILTER_HOOK /* tests/emul only: the arithmetic-policy experiment of docs/DSP_SPEC.md ("an fp32 filter kind?") */
  if (groove_emul_f32_filter) return groove_emul_lp24_step_f32(s, c, x, groove_emul_f32_filter);
#endif
  const double bx = c.b0a * x;
  const double y1 = bx + s.s0;
  s.s0 = fma(c.a1a, y1, 2.0 * bx + s.s1);
  s.s1 = fma(c.a2a, y1, bx);
  const double by = c.b0b * y1;
  const double y2 = by + s.s2;
  s.s2 = fma(c.a1b, y2, 2.0 * by + s.s3);
  s.s3 = fma(c.a2b, y2, by);
  return y2;
#endif
}

// ------------------------------------------------------------------ 24 dB low-pass, fp32 recurrence (round 5)
// The same two transposed-direct-form-II sections with fp32 coefficients and fp32 state: ten fp32 operations a frame instead of
// ten f64 ones, no widening of the coefficients on a retuning frame (four conversions and four f64 operations), no conversion of
// the input and the output.  Only for voices whose filter stays away from z = +1 AND z = -1 over its whole cutoff range
// (WF_FILTER_F32: the host measures it when the bank is uploaded, derive.h welsh_filter_f32_ok), and only in the per-kind serial
// kernels of big banks (kernels.h welsh_block<..., F32OK>): every other form keeps the f64 recurrence of DESIGN.md section 4.
// docs/DSP_SPEC.md "An fp32 filter kind" has the measurements that decided it.
struct Lp24CoefF { float b0a, a1a, a2a, b0b, a1b, a2b; };
struct Lp24StateF { float s0, s1, s2, s3; };
GROOVE_HD Lp24CoefF lp24_coeff_from_t(const Lp24Consts& c, float t, bool hi) { // lp24_coefd_from_t's quotients, finished in fp32
  const float T2 = t * t;
  const float dta = c.d1 * t, dtb = c.d3 * t;
  Lp24CoefF d;
  if (!hi) {
    const float ia = fast_rcp(c.c0 + dta + T2), ib = fast_rcp(c.c2 + dtb + T2);
    const float b0a = T2 * ia, q2a = (dta + dta) * ia, b0b = T2 * ib, q2b = (dtb + dtb) * ib;
    d.b0a = b0a; d.a1a = fmaf(-4.0f, b0a, 2.0f - q2a); d.a2a = q2a - 1.0f;
    d.b0b = b0b; d.a1b = fmaf(-4.0f, b0b, 2.0f - q2b); d.a2b = q2b - 1.0f;
  } else {
    const float Pa = c.c0 * T2, Pb = c.c2 * T2;
    const float ia = fast_rcp(1.0f + dta + Pa), ib = fast_rcp(1.0f + dtb + Pb);
    const float q2a = (dta + dta) * ia, pa = Pa * ia, q2b = (dtb + dtb) * ib, pb = Pb * ib;
    d.b0a = ia; d.a1a = fmaf(4.0f, pa, q2a - 2.0f); d.a2a = q2a - 1.0f;
    d.b0b = ib; d.a1b = fmaf(4.0f, pb, q2b - 2.0f); d.a2b = q2b - 1.0f;
  }
  return d;
}
GROOVE_HD Lp24CoefF lp24_coeff_from_fc(const Lp24Consts& c, float fc, float pi_over_sr, float fc_max) {
  bool hi;
  const float t = lp24_t_from_fc(fc, pi_over_sr, fc_max, hi);
  return lp24_coeff_from_t(c, t, hi);
}
GROOVE_HD Lp24CoefF lp24_coeff_from_pct(const Lp24Consts& c, float pct, const RenderConsts& rc) {
  bool hi;
  const float t = lp24_t_from_pct(pct, rc, hi);
  return lp24_coeff_from_t(c, t, hi);
}
// SCALAR_COEF: the coefficients are wave-uniform values the caller keeps in SGPRs (device only; static kinds).
template <bool SCALAR_COEF = false>
GROOVE_HD float lp24_step_f32(Lp24StateF& s, const Lp24CoefF& c, float x) { // lp24_step's ten operations
#if defined(__HIP_DEVICE_COMPILE__)
  // Written out for the same reason as the f64 step: left to itself the compiler picks the two-operand v_fmac_f32 for the state
  // updates and copies the four loop-carried state values (and two temporaries) back with six v_mov_b32 every frame — 6 of 16
  // instructions.  Three-operand v_fma_f32 updates each state value in place.  Same operations, same order, same roundings as the C.
  float bx, y1, t, by, y2, u;
  if constexpr (SCALAR_COEF) {
    asm("v_mul_f32 %[bx], %[b0a], %[x]\n\t"
        "v_add_f32 %[y1], %[bx], %[s0]\n\t"
        "v_fma_f32 %[t], %[bx], 2.0, %[s1]\n\t"
        "v_fma_f32 %[s0], %[a1a], %[y1], %[t]\n\t"
        "v_fma_f32 %[s1], %[a2a], %[y1], %[bx]\n\t"
        "v_mul_f32 %[by], %[b0b], %[y1]\n\t"
        "v_add_f32 %[y2], %[by], %[s2]\n\t"
        "v_fma_f32 %[u], %[by], 2.0, %[s3]\n\t"
        "v_fma_f32 %[s2], %[a1b], %[y2], %[u]\n\t"
        "v_fma_f32 %[s3], %[a2b], %[y2], %[by]"
        : [s0] "+v"(s.s0), [s1] "+v"(s.s1), [s2] "+v"(s.s2), [s3] "+v"(s.s3), [bx] "=&v"(bx), [y1] "=&v"(y1), [t] "=&v"(t),
          [by] "=&v"(by), [y2] "=&v"(y2), [u] "=&v"(u)
        : [x] "v"(x), [b0a] "s"(c.b0a), [a1a] "s"(c.a1a), [a2a] "s"(c.a2a), [b0b] "s"(c.b0b), [a1b] "s"(c.a1b), [a2b] "s"(c.a2b));
  } else {
    asm("v_mul_f32 %[bx], %[b0a], %[x]\n\t"
        "v_add_f32 %[y1], %[bx], %[s0]\n\t"
        "v_fma_f32 %[t], %[bx], 2.0, %[s1]\n\t"
        "v_fma_f32 %[s0], %[a1a], %[y1], %[t]\n\t"
        "v_fma_f32 %[s1], %[a2a], %[y1], %[bx]\n\t"
        "v_mul_f32 %[by], %[b0b], %[y1]\n\t"
        "v_add_f32 %[y2], %[by], %[s2]\n\t"
        "v_fma_f32 %[u], %[by], 2.0, %[s3]\n\t"
        "v_fma_f32 %[s2], %[a1b], %[y2], %[u]\n\t"
        "v_fma_f32 %[s3], %[a2b], %[y2], %[by]"
        : [s0] "+v"(s.s0), [s1] "+v"(s.s1), [s2] "+v"(s.s2), [s3] "+v"(s.s3), [bx] "=&v"(bx), [y1] "=&v"(y1), [t] "=&v"(t),
          [by] "=&v"(by), [y2] "=&v"(y2), [u] "=&v"(u)
        : [x] "v"(x), [b0a] "v"(c.b0a), [a1a] "v"(c.a1a), [a2a] "v"(c.a2a), [b0b] "v"(c.b0b), [a1b] "v"(c.a1b), [a2b] "v"(c.a2b));
  }
  return y2;
#else
  const float bx = c.b0a * x;
  const float y1 = bx + s.s0;
  s.s0 = fmaf(c.a1a, y1, fmaf(2.0f, bx, s.s1));
  s.s1 = fmaf(c.a2a, y1, bx);
  const float by = c.b0b * y1;
  const float y2 = by + s.s2;
  s.s2 = fmaf(c.a1b, y2, fmaf(2.0f, by, s.s3));
  s.s3 = fmaf(c.a2b, y2, by);
  return y2;
#endif
}
// ------------------------------------------------------------------ WelshVoice (a5)
// Packed per-voice flags word.
enum : uint32_t {
  WF_O1_WAVE_SHIFT = 0, WF_O2_WAVE_SHIFT = 4, WF_LFO_WAVE_SHIFT = 8, WF_ROUTING_SHIFT = 12,
  WF_SYNC = 1u << 16, WF_RETUNE_ENV = 1u << 17, WF_O2_FIXED = 1u << 18,
  WF_LFO_SMOOTH = 1u << 19, // host promise: the f64 LFO may be advanced by recurrences (see welsh_frame)
  // What the LFO drives, decoded from the routing once on the host (derive_welsh) so that every per-frame
  // test is one bit: pitch- or pulse-width-like (the edge-moving routings) and which oscillators they
  // reach, amplitude, cutoff percent, passband ripple.
  WF_LFO_PITCH = 1u << 20, WF_LFO_PW = 1u << 21, WF_LFO_O1 = 1u << 22, WF_LFO_O2 = 1u << 23,
  WF_LFO_AMP = 1u << 24, WF_LFO_CUTOFF = 1u << 25, WF_LFO_RESO = 1u << 26,
  WF_COEF_WIDE = 1u << 28, // host (derive.h): a RETUNED filter whose section constant c is below 1/512 (ripple above ~3.8): its poles sit towards z = -1 at any cutoff above c's own, and the per-frame coefficients take the two-sided form (lp24_coefd_from_t's `wide`; lp24_coefd_from_fc's in the exact-f64 kind and the time-parallel form)
  WF_FILTER_F32 = 1u << 27 // host promise (derive.h welsh_filter_f32_ok): the fp32 filter recurrence stays within 2e-6 of the f64 one over this patch's cutoff range
};
GROOVE_HD uint32_t lfo_routing_bits(uint32_t routing) {
  switch (routing) {
    case GROOVE_LFO_AMPLITUDE: return WF_LFO_AMP;
    case GROOVE_LFO_PITCH: return WF_LFO_PITCH | WF_LFO_O1 | WF_LFO_O2;
    case GROOVE_LFO_PITCH_OSC2: return WF_LFO_PITCH | WF_LFO_O2;
    case GROOVE_LFO_PULSE_WIDTH: return WF_LFO_PW | WF_LFO_O1 | WF_LFO_O2;
    case GROOVE_LFO_PW_OSC1: return WF_LFO_PW | WF_LFO_O1;
    case GROOVE_LFO_PW_OSC2: return WF_LFO_PW | WF_LFO_O2;
    case GROOVE_LFO_FILTER_CUTOFF: return WF_LFO_CUTOFF;
    case GROOVE_LFO_CUTOFF_AMP: return WF_LFO_CUTOFF | WF_LFO_AMP;
    case GROOVE_LFO_RESONANCE: return WF_LFO_RESO;
    default: return 0u;
  }
}
struct WelshParams {
  uint32_t flags;
  float mix;
  uint64_t o1_duty64, o2_duty64;
  float o1_duty, o2_duty;
  uint64_t lfo_inc;
  float lfo_depth;
  float cutoff_hz; // static cutoff
  EnvParams amp, fil;
  Lp24Consts fc;   // filter constants
  float ripple;    // passband ripple r itself (resonance routing: the constants are recomputed from r (1 + l depth))
  float cutoff_start, cutoff_end;
  float gl, gr;    // dca gain * pan law, per channel
  // LFO recurrence constants (host, f64): with D = 2 pi lfo_inc / 2^64 the per-frame rotation is
  // sin' = sin + (rs cos - rk sin), cos' = cos - (rs sin + rk cos), rk = 1 - cos D = 2 sin^2(D/2), rs = sin D;
  // lfo_a = lfo_depth * ln 2 (pitch routing: 2^(l depth) = e^(l lfo_a)).
  double lfo_rk, lfo_rs, lfo_a;
};
struct WelshState {
  OscState o1, o2, lfo;
  uint64_t o1_inc, o2_inc; // base increments (set by note_on)
  EnvState amp, fil;
  Lp24StateD filt;
  uint32_t vflags; // VF_FIRST
  uint32_t pad_;
};
// Per-block scratch that lives in registers across frames but is not persisted.
struct WelshScratch {
  Lp24CoefD coef;  // current filter coefficients
  Lp24CoefF coef_f; Lp24StateF filt_f; // ... and the filter's state, in the fp32 form (welsh_frame<..., F32FILT>: WF_FILTER_F32 voices)
  float prev_pct;  // cutoff percent the coefficients were computed for (NaN = none)
  double ls, lc;   // LFO_F64_SMOOTH: LFO value of the previous frame (sine: sin), and cos of the sine LFO's angle
  double lm;       // LFO_F64_SMOOTH, pitch routing: 2^(ls * depth)
  float ta, tf;    // HOIST: the two envelopes' stage counters as floats, for the frames of one segment
  float ac1, ac2, fc1, fc2; // HOIST: 2 D / len and D / len^2 of the two envelopes' current stages (value = n (c1 - c2 n) + A)
};
GROOVE_HD bool welsh_retunes(const WelshParams& p) {
  return (p.flags & (WF_RETUNE_ENV | WF_LFO_CUTOFF | WF_LFO_RESO)) != 0;
}
// "this frame computes its own coefficients": the look-ahead flag (welsh_frame's `tab`) tested as what it is, a scalar.  Through an
// opaque asm, one per use: left to itself the compiler folds the tests of a frame into lane-mask booleans, inverts them through vector
// instructions and evaluates what they guard before selecting it away.
GROOVE_HD bool welsh_tab_off(uint32_t tab) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(tab));
#endif
  return tab == 0u;
}
// ... or as what the caller KNOWS it to be: TABS (welsh_frame's template word: bit 0 `tab` is on, bit 1 `ltab` is on) — the frames of a
// chunk whose flags are all up run in a loop of their own without the tests (kernels.h run_frames_segmented, `fast_frame`).
template <int TABS, int BIT>
GROOVE_HD bool welsh_tab_is_off(uint32_t tab) {
  if constexpr ((TABS & BIT) != 0) return false;
  else return welsh_tab_off(tab);
}
// The cutoff percent an envelope-retuned filter takes from its envelope's value (DSP_SPEC section 6: start + (1 - start) end env).
GROOVE_HD float welsh_env_cutoff_pct(const WelshParams& p, float fil_value) {
  return fmaf((1.0f - p.cutoff_start) * p.cutoff_end, fil_value, p.cutoff_start);
}
// ... and the one an LFO-swept filter takes from the LFO's value.
GROOVE_HD float welsh_lfo_cutoff_pct(const WelshParams& p, float lfo) { return p.cutoff_start * fmaf(lfo, p.lfo_depth, 1.0f); }
// Filter constants from the passband ripple r on the device (fp32): the resonance routing moves r every
// frame.  sinh / cosh from one exp; r stays below ~22 (denormalize_q(1) (1 + depth)), far from overflow.
GROOVE_HD Lp24Consts lp24_consts_from_ripple(float r) {
  const float e = fast_exp2(r * 1.4426950408889634f), ie = fast_rcp(e);
  const float sg = 0.5f * (e - ie), ch = 0.5f * (e + ie), cg = ch * ch;
  Lp24Consts c;
  c.c0 = fast_rcp(cg - 0.85355339059327376220f); c.d1 = c.c0 * sg * 1.84775906502257351226f;
  c.c2 = fast_rcp(cg - 0.14644660940672623780f); c.d3 = c.c2 * sg * 0.76536686473017954346f;
  return c;
}

// How the LFO is evaluated.  LFO_F32: promise that no lane routes the LFO to Pitch or PulseWidth,
// which removes the f64 LFO / 2^x / u64<->f64 path (and ~80 VGPRs).  LFO_F64: exact per-frame
// evaluation from the 64-bit phase (any waveform).  LFO_F64_SMOOTH: promise that every lane that
// routes to Pitch / PulseWidth carries WF_LFO_SMOOTH (any LFO waveform but noise; |lfo_a dl| <= 1.5e-3 per
// frame BETWEEN the edges of a square / pulse / sawtooth LFO, whose frames evaluate exactly — round 6): frame 0 of a render call evaluates exactly and seeds (ls, lc, lm); later frames advance
// the sine by one rotation (6 f64 ops instead of a 10-term polynomial + conversions) and the
// pitch factor by lm *= e^(lfo_a (l - ls)) with a 4-term series (instead of a 15-term one).  The
// recurrences are re-seeded every block, so their error stays a random walk of <= 255 steps of
// ~1e-16 (tests/test_emul_numerics.py checks the result against the oracle).
enum : int { LFO_F32 = 0, LFO_F64 = 1, LFO_F64_SMOOTH = 2 };
GROOVE_HD int welsh_lfo_mode(const WelshParams& p) {
  // the resonance routing (per-frame sinh / cosh) is only compiled into the exact-f64 retuned kind: rare, and
  // it keeps the other five kinds free of its code and registers
  if (p.flags & WF_LFO_RESO) return LFO_F64; // (WF_COEF_WIDE patches ran there too until round 6: lp24_coefd_from_t's `wide` form serves them in every kind now)
  if (!(p.flags & (WF_LFO_PITCH | WF_LFO_PW))) return LFO_F32;
  return (p.flags & WF_LFO_SMOOTH) ? LFO_F64_SMOOTH : LFO_F64;
}

// Oscillator CLASSES: what the code of an audio oscillator can be specialised on at compile time.
// With wave-uniform patches "which waveform" is a scalar switch taken on every frame, and on gfx950
// the scalar pipe (one instruction per ~1.9 ns per SIMD, docs/VALU_COSTS.md) then carries as much
// work as the vector pipe: fixing the two audio oscillators' waveforms took 34 % off a frame.  The
// uniform kernels therefore dispatch ONCE per block on (class of oscillator 1, class of
// oscillator 2) to a copy of the frame loop compiled for that pair.  OSC_ANY keeps the run-time
// switch (noise, none, debug constants, triangle-sine; and every per-lane / unspecialised use).
enum : int { OSC_ANY = 0, OSC_PULSE = 1, OSC_SAW = 2, OSC_TRIANGLE = 3, OSC_SINE = 4, OSC_CLASSES = 5 };
// The LFO has the same classes (its waveform, when its value is used) plus LFO_UNUSED: routed nowhere
// and not a noise LFO, so only its phase advances.
enum : int { LFO_UNUSED = OSC_CLASSES, LFO_CLASSES = OSC_CLASSES + 1 };
GROOVE_HD int osc_class_of(uint32_t waveform) {
  switch (waveform) {
    case GROOVE_WAVE_SQUARE:
    case GROOVE_WAVE_PULSE_WIDTH: return OSC_PULSE; // same code; a square is duty 0.5
    case GROOVE_WAVE_SAWTOOTH: return OSC_SAW;
    case GROOVE_WAVE_TRIANGLE: return OSC_TRIANGLE;
    case GROOVE_WAVE_SINE: return OSC_SINE;
    default: return OSC_ANY;
  }
}
GROOVE_HD int lfo_class_of(uint32_t waveform, uint32_t routing) {
  if (lfo_routing_bits(routing) == 0u) return waveform == GROOVE_WAVE_NOISE ? (int)OSC_ANY : (int)LFO_UNUSED;
  return osc_class_of(waveform);
}
// Which kernel instantiation a patch asks for (kernels.h "Workgroup KINDS"): the BASE KIND (LFO mode x retune, in cost order: F32 static,
// F32 retune, SMOOTH static, SMOOTH retune, exact-F64 static, exact-F64 retune) and the class triple of the block body inside it.  Host
// and device share the rule (groove_hip.hip welsh_upload_params; tests/emul and tools/library_proportions.py read it through emul.cpp).
struct WelshParams;
GROOVE_HD int welsh_base_kind(const WelshParams& p);
GROOVE_HD bool welsh_base_kind_specialised(int base_kind) { (void)base_kind; return true; } // (round 6: the exact-f64 kinds too; they kept OSC_ANY bodies until then)
GROOVE_HD void welsh_body_classes(const WelshParams& p, int base_kind, int& cl, int& c1, int& c2);
template <int CLS>
GROOVE_HD uint32_t osc_class_wave(uint32_t runtime_waveform) {
  return CLS == OSC_PULSE ? (uint32_t)GROOVE_WAVE_PULSE_WIDTH : CLS == OSC_SAW ? (uint32_t)GROOVE_WAVE_SAWTOOTH
       : CLS == OSC_TRIANGLE ? (uint32_t)GROOVE_WAVE_TRIANGLE : CLS == OSC_SINE ? (uint32_t)GROOVE_WAVE_SINE : runtime_waveform;
}

GROOVE_HD int welsh_base_kind(const WelshParams& p) {
  const int mode = welsh_lfo_mode(p);
  return (mode == LFO_F32 ? 0 : (mode == LFO_F64_SMOOTH ? 2 : 4)) + (welsh_retunes(p) ? 1 : 0);
}
GROOVE_HD void welsh_body_classes(const WelshParams& p, int base_kind, int& cl, int& c1, int& c2) {
  const bool spec = welsh_base_kind_specialised(base_kind);
  c1 = spec ? osc_class_of((p.flags >> WF_O1_WAVE_SHIFT) & 15u) : (int)OSC_ANY;
  c2 = spec ? osc_class_of((p.flags >> WF_O2_WAVE_SHIFT) & 15u) : (int)OSC_ANY;
  cl = spec ? lfo_class_of((p.flags >> WF_LFO_WAVE_SHIFT) & 15u, (p.flags >> WF_ROUTING_SHIFT) & 15u) : (int)OSC_ANY;
  // the smooth-f64 kernels carry the sine / triangle / any LFO copies only (a square or sawtooth LFO there runs the `any` copy, which has
  // the exact re-seed on the frame of an LFO edge); the F32 and the exact-f64 kernels carry all six
  if ((base_kind == 2 || base_kind == 3) && cl != OSC_SINE && cl != OSC_TRIANGLE) cl = OSC_ANY;
}

// One frame of one voice.  FIRST: this is frame 0 of a render call (the only frame on which
// VF_FIRST can be set).  RETUNE: false promises !welsh_retunes(p) for every lane, so the
// coefficients in `sc` are loop-invariant.  LFO_MODE: see above.  C1, C2, CL: promise that every
// lane's oscillator 1 / oscillator 2 / LFO is of that class (OSC_ANY promises nothing).
// SEGMENT: the caller runs this frame inside a boundary-free segment (welsh_segment_begin) of a voice
// that is not idle, so neither the envelopes' boundary checks nor the idle test are needed here.
// REST: promise that an audio oscillator of class OSC_ANY has none of the classed waveforms (the
// host gives every wave of a class-specialised kind the body of its own classes, so OSC_ANY there
// means none / noise / triangle-sine / the debug constants).  Its value is then one FMA with two
// block-invariant scalars instead of the full waveform switch, which as scalar control flow costs
// ~30 instructions per oscillator per frame — nothing at full occupancy, 15 % of a frame for a lone wave.
template <int C, bool REST>
GROOVE_HD float osc_value_classed(uint32_t w, uint64_t phase, uint64_t duty64, float noise_value) {
  if constexpr (C == OSC_ANY && REST) {
    const float k_noise = w == GROOVE_WAVE_NOISE ? 1.0f : 0.0f;
    const float k_const = w == GROOVE_WAVE_DEBUG_MAX ? 1.0f : (w == GROOVE_WAVE_DEBUG_MIN ? -1.0f : 0.0f);
    float v = fmaf(noise_value, k_noise, k_const); // noise_value is 0 unless the waveform is noise
    if (w == GROOVE_WAVE_TRIANGLE_SINE) v = osc_value(GROOVE_WAVE_TRIANGLE_SINE, phase, duty64, 0.0f);
    return v;
  } else {
    return osc_value(w, phase, duty64, noise_value);
  }
}
// The frame in three parts (welsh_frame is their composition; the role-split kernel of welsh_split.h runs them on
// different wavefronts):
//   FRONT  everything feed-forward — envelopes, LFO, the two oscillators and their mix (`sum`, the filter's input), the cutoff
//          percent (`pct`; `retune` = the filter is to be retuned to it) and the output gain `a` (amplitude envelope x LFO);
//          returns false when the voice is idle on this (checked, non-SEGMENT) frame: it then contributes zeros and the
//          filter does not run.  `lfo` is handed out for the resonance routing.
//   COEF   the filter coefficients for `pct` (kept in sc.coef; an unchanged percent leaves them standing);
//   BACK   the f64 filter recurrence on `sum`, the gain, the pan gains.
// `tab` (a wave-uniform 0 / 1 in an SGPR; HOIST frames of the uniform kernels only): this segment's filter coefficients come from the
// wave's look-ahead table (kernels.h "coefficient look-ahead"), so the frame neither evaluates the filter envelope nor derives a cutoff
// percent from it.  (An integer, not a bool: as a bool the flag lived in a lane mask and its negation went through two vector instructions.)
// AMPTAB (with `tab`): the table's entries carry the AMPLITUDE envelope's value of the frame too (`tab_amp`; the caller sets `tab` only
// where the live lanes agree on BOTH envelopes' stages): the same env_shape on the same counter, from the lane that filled the entry.
template <bool FIRST, bool RETUNE, int LFO_MODE = LFO_F64, int C1 = OSC_ANY, int C2 = OSC_ANY, int CL = OSC_ANY, bool SEGMENT = false, bool REST = false,
          bool HOIST = false, bool AMPTAB = false, int TABS = 0, bool NO_PCT = false>
GROOVE_HD bool welsh_frame_front(const WelshParams& p, WelshState& s, WelshScratch& sc, float& sum, float& a, float& pct, bool& retune, float& lfo, uint32_t tab = 0u,
                                 uint32_t ltab = 0u, double tab_mod = 0.0, float tab_lfo = 0.0f, float tab_amp = 0.0f) {
  static_assert(!HOIST || (SEGMENT && !FIRST), "hoisted counters belong to a segment");
  if (SEGMENT && HOIST) {
    // env_shape with the segment's constants (welsh_segment_start_hoisted): two operations per envelope and frame
    // (The amplitude envelope's value from the table was first tried with a flag and a branch of its own, before the frame loop ran in
    // chunks: 0.362 - 0.373 against 0.354 - 0.357 ms per block in one job.  Lost in that form; AMPTAB rides on `tab`.)
    if (!AMPTAB) { s.amp.value = env_shape(sc.ta, s.amp.A, sc.ac1, sc.ac2); sc.ta += 1.0f; }
    // NO_PCT (welsh_frame's retuned HOIST frames): the filter envelope, the cutoff percent and the coefficients are one block behind ONE
    // test of `tab` there, not three tests here and there (each test three scalar instructions and, on a table frame, a taken branch)
    if (NO_PCT) {
    } else if (welsh_tab_is_off<TABS, 1>(tab)) {
      if (AMPTAB) { s.amp.value = env_shape(sc.ta, s.amp.A, sc.ac1, sc.ac2); sc.ta += 1.0f; }
      s.fil.value = env_shape(sc.tf, s.fil.A, sc.fc1, sc.fc2); sc.tf += 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
      // keeps `tab` a scalar BRANCH: if-converted, a table frame still evaluated the envelope and the percent and selected them away
      asm volatile("" : "+v"(s.fil.value));
      if (AMPTAB) asm volatile("" : "+v"(s.amp.value));
#endif
    } else if (AMPTAB) {
      s.amp.value = tab_amp;
    }
  } else if (SEGMENT) {
    env_advance(s.amp);
    env_advance(s.fil);
  } else {
    env_tick(s.amp, p.amp);
    env_tick(s.fil, p.fil);
    if (s.amp.state == ENV_IDLE) return false;
  }
  const uint32_t w1 = osc_class_wave<C1>((p.flags >> WF_O1_WAVE_SHIFT) & 15u), w2 = osc_class_wave<C2>((p.flags >> WF_O2_WAVE_SHIFT) & 15u);
  // LFO class: an unused LFO has no routing; a classed LFO in a static-filter f32 kind can only be
  // routed to the amplitude (cutoff routing retunes, pitch / pulse width are other LFO modes)
  const uint32_t wl = CL == LFO_UNUSED ? (uint32_t)GROOVE_WAVE_NONE : osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u);
  // What the LFO drives (WF_LFO_* bits).  Three promises turn the bit tests into constants: an unused LFO
  // drives nothing; a classed LFO in a static-filter f32 kind can only be routed to the amplitude (cutoff and
  // resonance routings retune, pitch / pulse width are other LFO modes); LFO_F64_SMOOTH is only ever chosen
  // for an LFO routed to pitch or pulse width (welsh_lfo_mode).  The resonance routing exists in the exact-f64
  // retuned kind only.
  constexpr bool NO_LFO = CL == LFO_UNUSED;
  constexpr bool AMP_ONLY = !NO_LFO && CL != OSC_ANY && LFO_MODE == LFO_F32 && !RETUNE;
  constexpr bool EDGE_ONLY = LFO_MODE == LFO_F64_SMOOTH;
  constexpr bool RESO = LFO_MODE == LFO_F64 && RETUNE;
  const uint32_t fl = p.flags;
  const bool r_edge = NO_LFO || AMP_ONLY || LFO_MODE == LFO_F32 ? false : (EDGE_ONLY ? true : (fl & (WF_LFO_PITCH | WF_LFO_PW)) != 0);
  const bool r_amp = NO_LFO || EDGE_ONLY ? false : (AMP_ONLY ? true : (fl & WF_LFO_AMP) != 0);
  const bool r_cut = NO_LFO || EDGE_ONLY || AMP_ONLY || !RETUNE ? false : (fl & WF_LFO_CUTOFF) != 0;
  const bool r_res = RESO && !NO_LFO ? (fl & WF_LFO_RESO) != 0 : false;
  const bool first = FIRST && (s.vflags & VF_FIRST);
  if (FIRST) s.vflags = 0;

  // LFO.  `mod` = what it does to the oscillators' edges (r_edge): the factor on their increments (pitch routing) or LFO value x depth
  // (pulse-width routing).  `ltab` (a wave-uniform 0 / 1 in an SGPR, like `tab`; HOIST frames, not the exact-f64 kinds): the wave's
  // live voices share the LFO's phase, and the caller hands this frame's LFO in from the wave's look-ahead table (kernels.h "LFO
  // look-ahead": evaluated from the phase, lane = frame) — `mod` in the smooth-f64 kinds (evaluated exactly; the caller re-seeds the
  // recurrences after the segment), the fp32 value `lfo` in the F32 kinds (the same expression on the same phase: the same bits) — and
  // moves the phase once per segment.
  constexpr bool LTAB = HOIST && LFO_MODE != LFO_F64 && CL != LFO_UNUSED;
  float nzl = 0.0f;
  const uint64_t half = 0x8000000000000000ull;
  uint64_t inc1 = s.o1_inc, inc2 = s.o2_inc;
  uint64_t d1 = p.o1_duty64, d2 = p.o2_duty64;
  lfo = 0.0f;
  double mod = 0.0;
  if (!LTAB || welsh_tab_is_off<TABS, 2>(ltab)) {
    if (!first && !(HOIST && NO_LFO)) s.lfo.phase += p.lfo_inc;
    if (wl == GROOVE_WAVE_NOISE) nzl = noise_tick(s.lfo);
    if (r_edge) {
      constexpr bool SMOOTH = LFO_MODE == LFO_F64_SMOOTH;
      double l;
      if (SMOOTH && !FIRST && wl == GROOVE_WAVE_SINE) { // one rotation step (an idle voice never gets here: see above)
        l = sc.ls + fma(p.lfo_rs, sc.lc, -(p.lfo_rk * sc.ls));
        sc.lc = sc.lc - fma(p.lfo_rs, sc.ls, p.lfo_rk * sc.lc);
      } else {
        l = osc_value_f64(wl, s.lfo.phase, half, nzl);
        if (SMOOTH && FIRST && wl == GROOVE_WAVE_SINE) // cos(2 pi x) = sin(2 pi (x + 1/4))
          sc.lc = sin_turns_folded_f64((double)fold_quarter64((int64_t)(s.lfo.phase + 0x4000000000000000ull)) * 5.42101086242752217004e-20);
      }
      if (fl & WF_LFO_PITCH) {
        double m;
        if (SMOOTH && !FIRST) {
          // an LFO with EDGES (square, pulse, sawtooth: class OSC_ANY in the smooth kinds) is smooth between them; on the frame of an
          // edge the factor is evaluated exactly, which re-seeds the recurrence (a square's factor is then constant, bit for bit:
          // e^0 = 1).  Until round 6 such patches ran in the exact-f64 kind, a 15-term series on every frame.
          const double dx = (l - sc.ls) * p.lfo_a;
          if (CL == OSC_ANY && fabs(dx) > 1.5e-3) m = exp2_small_f64(l * (double)p.lfo_depth);
          else m = sc.lm * exp_tiny_f64(dx);
        }
        else m = exp2_small_f64(l * (double)p.lfo_depth);
        if (SMOOTH) sc.lm = m;
        mod = m;
      } else {
        mod = l * (double)p.lfo_depth;
      }
      if (SMOOTH) sc.ls = l;
      lfo = (float)l;
#if defined(__HIP_DEVICE_COMPILE__)
      if (LTAB) asm volatile("" : "+v"(mod)); // keeps `ltab` a scalar BRANCH (as `tab` below)
#endif
    } else if (r_amp || r_cut || r_res) {
      lfo = osc_value(wl, s.lfo.phase, half, nzl);
#if defined(__HIP_DEVICE_COMPILE__)
      if (LTAB) asm volatile("" : "+v"(lfo));
#endif
    }
  } else {
    mod = tab_mod;
    lfo = tab_lfo;
  }
  if (r_edge) {
    if (fl & WF_LFO_PITCH) {
      if (fl & WF_LFO_O1) inc1 = f64_to_u64((double)inc1 * mod);
      if (fl & WF_LFO_O2) inc2 = f64_to_u64((double)inc2 * mod); // fm applies to a fixed-frequency osc too
    } else {
      if (fl & WF_LFO_O1) d1 = f64_to_u64(clamp01d((double)p.o1_duty * (1.0 + mod)) * 18446744073709549568.0);
      if (fl & WF_LFO_O2) d2 = f64_to_u64(clamp01d((double)p.o2_duty * (1.0 + mod)) * 18446744073709549568.0);
    }
  }

  // oscillators (+ hard sync): carry out of the 64-bit add = osc 1 wrapped
  bool wrapped = false;
  if (!first) {
    const uint64_t np = s.o1.phase + inc1;
    wrapped = np < inc1; // carry out of the add (== np < the old phase): compared with the increment, so that the add is in place
    s.o1.phase = np;
  }
  float nz1 = 0.0f, nz2 = 0.0f;
  if (w1 == GROOVE_WAVE_NOISE) nz1 = noise_tick(s.o1);
  {
    uint64_t ph2 = s.o2.phase;
    if (!first) ph2 += inc2;
    if (fl & WF_SYNC) { // wave-uniform in the uniform kernels: a scalar branch that five patches in six never take
      if (wrapped) ph2 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(ph2)); // keeps this a branch: if-converted it is a 64-bit compare and three selects on every frame
#endif
    }
    s.o2.phase = ph2;
  }
  if (w2 == GROOVE_WAVE_NOISE) nz2 = noise_tick(s.o2);
  const float v1 = osc_value_classed<C1, REST>(w1, s.o1.phase, d1, nz1);
  const float v2 = osc_value_classed<C2, REST>(w2, s.o2.phase, d2, nz2);
  sum = fmaf(v1, p.mix, v2 * (1.0f - p.mix));

  // filter cutoff
  retune = false;
  pct = 0.0f;
  if (RETUNE && !NO_PCT && welsh_tab_is_off<TABS, 1>(tab)) {
    // (an unused LFO cannot drive the cutoff: in a retuned kind the envelope must)
    if (CL == LFO_UNUSED || (p.flags & WF_RETUNE_ENV)) {
      pct = welsh_env_cutoff_pct(p, s.fil.value);
      retune = true;
    } else if (r_cut) {
      pct = welsh_lfo_cutoff_pct(p, lfo);
      retune = true;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (HOIST) asm volatile("" : "+v"(pct)); // (as above)
#endif
  }
  a = s.amp.value;
  if (r_amp) a *= fmaf(lfo, p.lfo_depth, 1.0f);
  return true;
}
// FRONT once more, in two halves, for the four-role kernel of welsh_split.h, which runs them on different wavefronts.
// The statements are welsh_frame_front's (not exact-f64 LFO kinds: LFO_F32 and LFO_F64_SMOOTH only), each on the same values
// in the same order within its half, so CTL + OSC give FRONT's results bit for bit (tests/emul: test_emul_numerics.py;
// tests/test_gpu_split.py on the device):
//   CTL  the two envelopes and the LFO -> the gain `a`, the cutoff percent, and `mod`: what the LFO does to the oscillators'
//        edges (pitch routing: the factor on their increments; pulse-width routing: LFO value x depth).  Owns the state's
//        envelope, LFO and flag words.  `first` = the voice's first tick after a note-on (FIRST frames only).
//   OSC  the two oscillators, hard sync, their mix.  Owns the oscillators' words; reads the base increments.
template <bool FIRST, bool RETUNE, int LFO_MODE, int CL, bool SEGMENT = false, bool HOIST = false>
GROOVE_HD bool welsh_frame_ctl(const WelshParams& p, WelshState& s, WelshScratch& sc, float& a, float& pct, bool& retune, double& mod, bool& first) {
  static_assert(LFO_MODE != LFO_F64, "the exact-f64 kinds keep the whole frame on one wavefront");
  static_assert(!HOIST || (SEGMENT && !FIRST), "hoisted counters belong to a segment");
  if (SEGMENT && HOIST) {
    // env_shape with the segment's constants (welsh_segment_start_hoisted): two operations per envelope and frame
    s.amp.value = env_shape(sc.ta, s.amp.A, sc.ac1, sc.ac2);
    s.fil.value = env_shape(sc.tf, s.fil.A, sc.fc1, sc.fc2);
    sc.ta += 1.0f; sc.tf += 1.0f;
  } else if (SEGMENT) {
    env_advance(s.amp);
    env_advance(s.fil);
  } else {
    env_tick(s.amp, p.amp);
    env_tick(s.fil, p.fil);
    if (s.amp.state == ENV_IDLE) return false;
  }
  const uint32_t wl = CL == LFO_UNUSED ? (uint32_t)GROOVE_WAVE_NONE : osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u);
  constexpr bool NO_LFO = CL == LFO_UNUSED;
  constexpr bool AMP_ONLY = !NO_LFO && CL != OSC_ANY && LFO_MODE == LFO_F32 && !RETUNE;
  constexpr bool EDGE_ONLY = LFO_MODE == LFO_F64_SMOOTH;
  const uint32_t fl = p.flags;
  const bool r_edge = EDGE_ONLY && !NO_LFO;
  const bool r_amp = NO_LFO || EDGE_ONLY ? false : (AMP_ONLY ? true : (fl & WF_LFO_AMP) != 0);
  const bool r_cut = NO_LFO || EDGE_ONLY || AMP_ONLY || !RETUNE ? false : (fl & WF_LFO_CUTOFF) != 0;
  first = FIRST && (s.vflags & VF_FIRST);
  if (FIRST) s.vflags = 0;
  if (!first && !(HOIST && NO_LFO)) s.lfo.phase += p.lfo_inc;
  float nzl = 0.0f;
  if (wl == GROOVE_WAVE_NOISE) nzl = noise_tick(s.lfo);
  const uint64_t half = 0x8000000000000000ull;
  float lfo = 0.0f;
  mod = 0.0;
  if (r_edge) {
    double l;
    if (!FIRST && wl == GROOVE_WAVE_SINE) {
      l = sc.ls + fma(p.lfo_rs, sc.lc, -(p.lfo_rk * sc.ls));
      sc.lc = sc.lc - fma(p.lfo_rs, sc.ls, p.lfo_rk * sc.lc);
    } else {
      l = osc_value_f64(wl, s.lfo.phase, half, nzl);
      if (FIRST && wl == GROOVE_WAVE_SINE)
        sc.lc = sin_turns_folded_f64((double)fold_quarter64((int64_t)(s.lfo.phase + 0x4000000000000000ull)) * 5.42101086242752217004e-20);
    }
    if (fl & WF_LFO_PITCH) {
      double m;
      if (!FIRST) { // (welsh_frame_front's statements: an LFO with edges is re-seeded exactly on the frame of an edge)
        const double dx = (l - sc.ls) * p.lfo_a;
        if (CL == OSC_ANY && fabs(dx) > 1.5e-3) m = exp2_small_f64(l * (double)p.lfo_depth);
        else m = sc.lm * exp_tiny_f64(dx);
      }
      else m = exp2_small_f64(l * (double)p.lfo_depth);
      sc.lm = m;
      mod = m;
    } else {
      mod = l * (double)p.lfo_depth;
    }
    sc.ls = l;
    lfo = (float)l;
  } else if (r_amp || r_cut) {
    lfo = osc_value(wl, s.lfo.phase, half, nzl);
  }
  retune = false;
  pct = 0.0f;
  if (RETUNE) {
    if (CL == LFO_UNUSED || (p.flags & WF_RETUNE_ENV)) {
      pct = fmaf((1.0f - p.cutoff_start) * p.cutoff_end, s.fil.value, p.cutoff_start);
      retune = true;
    } else if (r_cut) {
      pct = p.cutoff_start * fmaf(lfo, p.lfo_depth, 1.0f);
      retune = true;
    }
  }
  a = s.amp.value;
  if (r_amp) a *= fmaf(lfo, p.lfo_depth, 1.0f);
  return true;
}
// (`edge`: the LFO reaches the oscillators — every wavefront of an LFO_F64_SMOOTH kind whose LFO class is not LFO_UNUSED.)
template <int LFO_MODE, int C1, int C2, bool REST>
GROOVE_HD float welsh_frame_osc(const WelshParams& p, WelshState& s, bool edge, double mod, bool first) {
  const uint32_t w1 = osc_class_wave<C1>((p.flags >> WF_O1_WAVE_SHIFT) & 15u), w2 = osc_class_wave<C2>((p.flags >> WF_O2_WAVE_SHIFT) & 15u);
  const uint32_t fl = p.flags;
  uint64_t inc1 = s.o1_inc, inc2 = s.o2_inc;
  uint64_t d1 = p.o1_duty64, d2 = p.o2_duty64;
  if (LFO_MODE == LFO_F64_SMOOTH && edge) {
    if (fl & WF_LFO_PITCH) {
      if (fl & WF_LFO_O1) inc1 = f64_to_u64((double)inc1 * mod);
      if (fl & WF_LFO_O2) inc2 = f64_to_u64((double)inc2 * mod);
    } else {
      if (fl & WF_LFO_O1) d1 = f64_to_u64(clamp01d((double)p.o1_duty * (1.0 + mod)) * 18446744073709549568.0);
      if (fl & WF_LFO_O2) d2 = f64_to_u64(clamp01d((double)p.o2_duty * (1.0 + mod)) * 18446744073709549568.0);
    }
  }
  bool wrapped = false;
  if (!first) {
    const uint64_t np = s.o1.phase + inc1;
    wrapped = np < inc1; // carry out of the add (== np < the old phase): compared with the increment, so that the add is in place
    s.o1.phase = np;
  }
  float nz1 = 0.0f, nz2 = 0.0f;
  if (w1 == GROOVE_WAVE_NOISE) nz1 = noise_tick(s.o1);
  {
    uint64_t ph2 = s.o2.phase;
    if (!first) ph2 += inc2;
    if (fl & WF_SYNC) {
      if (wrapped) ph2 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(ph2));
#endif
    }
    s.o2.phase = ph2;
  }
  if (w2 == GROOVE_WAVE_NOISE) nz2 = noise_tick(s.o2);
  const float v1 = osc_value_classed<C1, REST>(w1, s.o1.phase, d1, nz1);
  const float v2 = osc_value_classed<C2, REST>(w2, s.o2.phase, d2, nz2);
  return fmaf(v1, p.mix, v2 * (1.0f - p.mix));
}
template <bool RETUNE, int LFO_MODE, int CL>
GROOVE_HD void welsh_frame_coef(const WelshParams& p, const RenderConsts& rc, WelshScratch& sc, float pct, bool retune, float lfo) {
  if (RETUNE) {
    constexpr bool RESO = LFO_MODE == LFO_F64 && RETUNE;
    const bool r_res = RESO && CL != LFO_UNUSED ? (p.flags & WF_LFO_RESO) != 0 : false;
    if (RESO && r_res) { // the ripple moves every frame: constants and coefficients are recomputed
      const Lp24Consts c = lp24_consts_from_ripple(p.ripple * fmaf(lfo, p.lfo_depth, 1.0f));
      const float fc = retune ? 25.0f * fast_exp2(clamp01f(pct) * 9.6438561897747244f) : p.cutoff_hz;
      sc.coef = lp24_coefd_from_fc(c, fc, rc.pi_over_sr, rc.fc_max);
    } else if (retune && pct != sc.prev_pct) { // unchanged percent (e.g. envelope plateau): coefficients stand
      sc.coef = lp24_coefd_from_pct(p.fc, pct, rc, (p.flags & WF_COEF_WIDE) != 0); // (a scalar branch in the uniform kernels)
      sc.prev_pct = pct;
    }
  }
}
template <bool SCALAR_COEF>
GROOVE_HD void welsh_frame_back(const WelshParams& p, Lp24StateD& filt, const Lp24CoefD& coef, float sum, float a, float& L, float& R) {
  const float y = (float)lp24_step<SCALAR_COEF>(filt, coef, (double)sum); // uniform static kinds: coefficients in SGPRs
  const float m = y * a;
  L = m * p.gl;
  R = m * p.gr;
}
// HOIST (with SEGMENT): the caller keeps the envelopes' stage counters for the segment — as floats in sc.ta /
// sc.tf, which this frame reads and bumps (exact below 2^24 frames per stage), the integer counters moving once,
// after the segment (welsh_segment_end) — and an unused LFO's phase moves there too: two conversions, two integer
// adds and a 64-bit add less on every frame.
template <bool FIRST, bool RETUNE, int LFO_MODE = LFO_F64, int C1 = OSC_ANY, int C2 = OSC_ANY, int CL = OSC_ANY, bool SEGMENT = false, bool REST = false,
          bool HOIST = false, bool F32FILT = false, bool AMPTAB = false, int TABS = 0>
GROOVE_HD void welsh_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc,
                           WelshScratch& sc, float& L, float& R, uint32_t tab = 0u, uint32_t ltab = 0u, double tab_mod = 0.0, float tab_lfo = 0.0f, float tab_amp = 0.0f) {
  static_assert(!(F32FILT && LFO_MODE == LFO_F64), "the exact-f64 kinds (resonance routing) keep the f64 filter");
  float sum, a, pct, lfo;
  bool retune;
  // (tab: the caller has put this frame's coefficients into sc.coef / sc.coef_f already — kernels.h "coefficient look-ahead")
  constexpr bool LATE = HOIST && RETUNE && LFO_MODE != LFO_F64 && !AMPTAB; // filter envelope -> percent -> coefficients in one block, below
  if (!welsh_frame_front<FIRST, RETUNE, LFO_MODE, C1, C2, CL, SEGMENT, REST, HOIST, AMPTAB, TABS, LATE>(p, s, sc, sum, a, pct, retune, lfo, tab, ltab, tab_mod, tab_lfo, tab_amp)) { L = 0.0f; R = 0.0f; return; }
  if constexpr (LATE) {
    if (welsh_tab_is_off<TABS, 1>(tab)) {
      // welsh_frame_front's statements, in its order: the filter envelope's value of the frame, then the percent from it — or from the LFO
      s.fil.value = env_shape(sc.tf, s.fil.A, sc.fc1, sc.fc2); sc.tf += 1.0f;
      if (CL == LFO_UNUSED || (p.flags & WF_RETUNE_ENV)) { pct = welsh_env_cutoff_pct(p, s.fil.value); retune = true; }
      else if (LFO_MODE == LFO_F32 && (p.flags & WF_LFO_CUTOFF)) { pct = welsh_lfo_cutoff_pct(p, lfo); retune = true; }
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(pct)); // keeps `tab` a scalar BRANCH: if-converted, a table frame still evaluated all this and selected it away
#endif
      if constexpr (F32FILT) { if (retune && pct != sc.prev_pct) { sc.coef_f = lp24_coeff_from_pct(p.fc, pct, rc); sc.prev_pct = pct; } }
      else welsh_frame_coef<RETUNE, LFO_MODE, CL>(p, rc, sc, pct, retune, lfo);
    }
    if constexpr (F32FILT) {
      const float m = lp24_step_f32<false>(sc.filt_f, sc.coef_f, sum) * a;
      L = m * p.gl; R = m * p.gr;
    } else {
      welsh_frame_back<false>(p, s.filt, sc.coef, sum, a, L, R);
    }
    return;
  }
  if constexpr (F32FILT) { // sc.coef_f / sc.filt_f were set by welsh_scratch_f32_begin; the caller hands the state back with welsh_scratch_f32_end
    if (RETUNE && welsh_tab_is_off<TABS, 1>(tab)) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (HOIST) asm volatile("" : "+v"(pct)); // (the comparison below stays inside this branch)
#endif
      if (retune && pct != sc.prev_pct) { sc.coef_f = lp24_coeff_from_pct(p.fc, pct, rc); sc.prev_pct = pct; }
    }
    const float m = lp24_step_f32<SEGMENT && !RETUNE>(sc.filt_f, sc.coef_f, sum) * a; // uniform static kinds: coefficients in SGPRs
    L = m * p.gl; R = m * p.gr;
  } else {
    if (welsh_tab_is_off<TABS, 1>(tab)) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (HOIST && RETUNE) asm volatile("" : "+v"(pct));
#endif
      welsh_frame_coef<RETUNE, LFO_MODE, CL>(p, rc, sc, pct, retune, lfo);
    }
    welsh_frame_back<SEGMENT && !RETUNE>(p, s.filt, sc.coef, sum, a, L, R);
  }
}
// The fp32 filter form's block bracket: the state is PERSISTED in the f64 fields of WelshState (which then hold fp32 values
// exactly), so every other kernel form can pick the voice up with its f64 recurrence.
GROOVE_HD void welsh_scratch_f32_begin(const WelshParams& p, const WelshState& s, const RenderConsts& rc, WelshScratch& sc) {
  sc.coef_f = lp24_coeff_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max);
  sc.filt_f = Lp24StateF{(float)s.filt.s0, (float)s.filt.s1, (float)s.filt.s2, (float)s.filt.s3};
}
GROOVE_HD void welsh_scratch_f32_end(WelshState& s, const WelshScratch& sc) {
  s.filt.s0 = (double)sc.filt_f.s0; s.filt.s1 = (double)sc.filt_f.s1; s.filt.s2 = (double)sc.filt_f.s2; s.filt.s3 = (double)sc.filt_f.s3;
}
// Segments.  Between two envelope stage boundaries nothing about a voice's control flow changes: the
// boundary checks of both envelopes and the idle test can be made once, and the frames up to the next
// boundary run without them.  welsh_segment_begin handles any boundary due now and returns how many
// frames THIS voice can run unchecked (>= 1); a wave takes the minimum over its lanes.  `live` = the
// voice sounds in this segment (its frames go through welsh_frame<..., SEGMENT = true>); an idle
// voice only advances its envelope counters (welsh_segment_idle_frame).  Frame for frame the
// operations on every piece of state are those of the checked form.
GROOVE_HD uint32_t welsh_segment_begin(const WelshParams& p, WelshState& s, bool& live) {
  env_boundary(s.amp, p.amp);
  env_boundary(s.fil, p.fil);
  live = s.amp.state != ENV_IDLE;
  const uint32_t a = env_frames_to_boundary(s.amp), f = env_frames_to_boundary(s.fil);
  return a < f ? a : f;
}
GROOVE_HD void welsh_segment_idle_frame(WelshState& s) {
  env_advance(s.amp);
  env_advance(s.fil);
}
// Hoisted form (welsh_frame<..., HOIST>): what a segment of `seg` frames leaves to do once.  Live voices had
// their envelope VALUES set by their frames; an idle voice's are what its last tick would have produced.
GROOVE_HD void welsh_segment_start_hoisted(const WelshState& s, WelshScratch& sc) {
  sc.ta = (float)s.amp.n; sc.tf = (float)s.fil.n;
  env_shape_consts(s.amp.D, s.amp.inv_len, sc.ac1, sc.ac2);
  env_shape_consts(s.fil.D, s.fil.inv_len, sc.fc1, sc.fc2);
}
template <bool LFO_TOO>
GROOVE_HD void welsh_segment_end_hoisted(const WelshParams& p, WelshState& s, uint32_t seg, bool live) {
  s.amp.n += seg; s.fil.n += seg;
  if (!live) { s.amp.value = env_last_value_of(s.amp); s.fil.value = env_last_value_of(s.fil); }
  else if (LFO_TOO) s.lfo.phase += (uint64_t)seg * p.lfo_inc;
}
// LFO look-ahead (kernels.h) of the smooth-f64 kinds.  What the LFO does to the oscillators' edges on the frame whose LFO phase is
// `phase`, evaluated exactly — the exact-f64 kind's expressions (welsh_frame_front, LFO_MODE == LFO_F64) ...
template <int CL>
GROOVE_HD double welsh_lfo_mod_exact(const WelshParams& p, uint64_t phase) {
  const uint32_t wl = osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u);
  const double ld = osc_value_f64(wl, phase, 0x8000000000000000ull, 0.0f) * (double)p.lfo_depth; // (never a noise LFO: WF_LFO_SMOOTH)
  return (p.flags & WF_LFO_PITCH) ? exp2_small_f64(ld) : ld;
}
// (the F32 kinds' LFO value on the frame whose phase is `phase`: welsh_frame_front's expression; never a noise LFO — the caller checks)
template <int CL>
GROOVE_HD float welsh_lfo_value_f32(const WelshParams& p, uint64_t phase) {
  return osc_value(osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u), phase, 0x8000000000000000ull, 0.0f);
}
// ... and, after a segment whose frames took `mod` from the table and whose phase moved once, the recurrences' state as a FIRST frame
// seeds it: the frames of a later segment that has to run lane by lane continue from it.
template <int CL>
GROOVE_HD void welsh_lfo_reseed_smooth(const WelshParams& p, const WelshState& s, WelshScratch& sc) {
  const uint32_t wl = osc_class_wave<CL>((p.flags >> WF_LFO_WAVE_SHIFT) & 15u);
  const double l = osc_value_f64(wl, s.lfo.phase, 0x8000000000000000ull, 0.0f);
  if (wl == GROOVE_WAVE_SINE)
    sc.lc = sin_turns_folded_f64((double)fold_quarter64((int64_t)(s.lfo.phase + 0x4000000000000000ull)) * 5.42101086242752217004e-20);
  if (p.flags & WF_LFO_PITCH) sc.lm = exp2_small_f64(l * (double)p.lfo_depth);
  sc.ls = l;
}
GROOVE_HD WelshScratch welsh_scratch_init(const WelshParams& p, const RenderConsts& rc) {
  WelshScratch sc;
  sc.coef = lp24_coefd_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max);
  sc.prev_pct = __builtin_nanf("");
  sc.ls = 0.0; sc.lc = 1.0; sc.lm = 1.0;
  return sc;
}

// ------------------------------------------------------------------ FmVoice (a6)
struct FmParams {
  double depth_beta; // depth * beta
  EnvParams cenv, menv;
  float gl, gr;
};
struct FmState {
  OscState carrier, modulator;
  uint64_t c_inc, m_inc;
  EnvState cenv, menv;
  uint32_t vflags, pad_;
};
template <bool FIRST>
GROOVE_HD void fm_frame(const FmParams& p, FmState& s, float& L, float& R) {
  env_tick(s.cenv, p.cenv);
  env_tick(s.menv, p.menv);
  if (s.cenv.state == ENV_IDLE) { L = 0.0f; R = 0.0f; return; }
  const bool first = FIRST && (s.vflags & VF_FIRST);
  if (FIRST) s.vflags = 0;
  if (!first) s.modulator.phase += s.m_inc;
  const double mv = osc_value_f64(GROOVE_WAVE_SINE, s.modulator.phase, 0, 0.0f);
  const double lfm = mv * (double)s.menv.value * p.depth_beta;
  // carrier delta = base * (2^0 + lfm); may go negative (through-zero FM)
  const double turns = (double)s.c_inc * 5.42101086242752217004e-20 * (1.0 + lfm);
  if (!first) s.carrier.phase += turns_to_inc(turns);
  const float cv = osc_value(GROOVE_WAVE_SINE, s.carrier.phase, 0, 0.0f);
  const float m = cv * s.cenv.value;
  L = m * p.gl;
  R = m * p.gr;
}

// ------------------------------------------------------------------ SamplerVoice (a7)
// idx / step are Q20.44 fixed point (buffers up to 2^20 frames, 2^-44 step resolution).
struct SamplerParams {
  uint32_t offset, length; // frames into the shared bank
  double root_hz;          // <= 0: drumkit (step 1)
  float gain;
  uint32_t one_shot;
};
struct SamplerState {
  uint64_t idx, step;
  uint32_t playing;
  uint32_t pad_;
};
GROOVE_HD float sampler_frame(const SamplerParams& p, SamplerState& s, const float* bank) {
  if (!s.playing) return 0.0f;
  const uint32_t i = (uint32_t)(s.idx >> 44);
  if (i >= p.length) { s.playing = 0; return 0.0f; }
  const float v = bank[(size_t)p.offset + i] * p.gain;
  s.idx += s.step;
  return v;
}

// C consecutive frames at once (count <= C of them wanted), same results as `count` calls of
// sampler_frame.  The index only moves forward, so "still inside the sample" is monotone over the
// chunk: all C fetches are issued from clamped addresses before any is used (the per-frame form
// pays one cache round trip per frame: the fetch is the only long-latency operation of a voice).
template <int C>
GROOVE_HD void sampler_chunk(const SamplerParams& p, SamplerState& s, const float* bank, uint32_t count, float (&v)[C]) {
  uint32_t valid = 0; // frames of the chunk that play
  float raw[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t i = (uint32_t)((s.idx + (uint64_t)k * s.step) >> 44);
    const bool ok = s.playing && (uint32_t)k < count && i < p.length;
    raw[k] = bank[(size_t)p.offset + (i < p.length ? i : p.length - 1)];
    valid += ok ? 1u : 0u;
    v[k] = ok ? 1.0f : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < C; ++k) v[k] = v[k] != 0.0f ? raw[k] * p.gain : 0.0f;
  s.idx += (uint64_t)valid * s.step;
  if (s.playing && valid < count) s.playing = 0; // ran off the end inside the chunk
}

// ------------------------------------------------------------------ effects (a8, a9)
GROOVE_HD float bitcrush(float x, uint32_t bits) {
  float ax = fabsf(x) * 32767.0f;
  if (!(ax < 2147483648.0f)) ax = 2147483520.0f;
  uint32_t q = (uint32_t)ax;
  q = (q >> bits) << bits;
  const float y = (float)q * (1.0f / 32767.0f);
  return copysignf(y, x);
}
GROOVE_HD float limiter(float x, float mn, float mx) {
  return copysignf(fminf(fmaxf(fabsf(x), mn), mx), x);
}
GROOVE_HD float compressor(float x, float threshold, float ratio) {
  float a = fabsf(x);
  if (a > threshold) a = fmaf(a - threshold, ratio, threshold);
  return copysignf(a, x);
}
// Biquad Direct Form 1 (doc/Audio-EQ-Cookbook.txt Eq 4), f64 state and coefficients.
struct BiquadCoefD { double b0, b1, b2, a1, a2; };
struct BiquadStateD { double x1, x2, y1, y2; };
GROOVE_HD double biquad_step(BiquadStateD& s, const BiquadCoefD& c, double x) {
  const double y = c.b0 * x + c.b1 * s.x1 + c.b2 * s.x2 - c.a1 * s.y1 - c.a2 * s.y2;
  s.x2 = s.x1; s.x1 = x; s.y2 = s.y1; s.y1 = y;
  return y;
}

} // namespace groove
