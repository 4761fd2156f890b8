// derive.h — host-side (f64) derivation of the packed per-lane device parameters from
// the public `*Params` structs, i.e. what `Foo::new_with(&FooParams)` +
// `Configurable::update_sample_rate` do in the reference
// (/root/reference/settings/src/instruments.rs:71-92, orchestrator.rs:125-127).
// Host only; shared by the C ABI implementation and the tests/emul numerics harness.
#pragma once
#include "dsp_core.h"

namespace groove {

inline uint32_t frames_f64(double len) { // (dsp_core.h env_frames, in f64: the stage lengths that never change are counted on the host)
  if (!(len > 0.0)) return 0u;
  double c = ceil(len * (1.0 - 1.0 / 65536.0));
  return c > 4.0e9 ? 4000000000u : (uint32_t)c;
}
inline double clamp01_h(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

inline EnvParams derive_env(const groove_envelope_params& e, double sr) {
  EnvParams o;
  const double sustain = clamp01_h(e.sustain);
  const double al = e.attack * sr * (1.0 - 0.0); // same expression as the oracle's enter(ATTACK, 0)
  const double dl = e.decay * sr * (1.0 - sustain);
  o.attack_len = (float)al; o.attack_N = frames_f64(al);
  o.decay_len = (float)dl; o.decay_N = frames_f64(dl);
  o.sustain = (float)sustain;
  o.release_len = (float)(e.release * sr);
  return o;
}
inline uint64_t duty_to_u64(double duty) {
  return (uint64_t)(clamp01_h(duty) * 18446744073709549568.0); // (2^64 - 2048): no overflow at 1.0
}
inline void pan_gains(double gain, double pan, float& gl, float& gr) {
  gl = (float)(gain * (1.0 - 0.25 * (pan + 1.0) * (pan + 1.0)));
  gr = (float)(gain * (1.0 - (0.5 * pan - 0.5) * (0.5 * pan - 0.5)));
}
inline Lp24Consts derive_lp24_consts(double ripple) {
  double sg = sinh(ripple), cg = cosh(ripple);
  cg *= cg;
  const double c0 = 1.0 / (cg - 0.85355339059327376220);
  const double c2 = 1.0 / (cg - 0.14644660940672623780);
  Lp24Consts c;
  c.c0 = (float)c0; c.d1 = (float)(c0 * sg * 1.84775906502257351226);
  c.c2 = (float)c2; c.d3 = (float)(c2 * sg * 0.76536686473017954346);
  return c;
}

// Cold per-voice values only the note-event kernel needs.
struct WelshCold { double tune1, tune2, fixed1, fixed2; };

inline WelshParams derive_welsh(const groove_welsh_params& p, double sr, WelshCold& cold) {
  WelshParams o{};
  const uint32_t w1 = p.oscillator_1.waveform & 15u, w2 = p.oscillator_2.waveform & 15u;
  o.flags = (w1 << WF_O1_WAVE_SHIFT) | (w2 << WF_O2_WAVE_SHIFT) |
            ((p.lfo_waveform & 15u) << WF_LFO_WAVE_SHIFT) | ((p.lfo_routing & 15u) << WF_ROUTING_SHIFT);
  o.flags |= lfo_routing_bits(p.lfo_routing & 15u);
  if (p.oscillator_2_sync) o.flags |= WF_SYNC;
  if (p.filter_cutoff_end != 0.0f) o.flags |= WF_RETUNE_ENV;
  if (p.oscillator_2.fixed_hz > 0.0) o.flags |= WF_O2_FIXED;
  o.mix = p.oscillator_mix;
  o.o1_duty = (w1 == GROOVE_WAVE_PULSE_WIDTH) ? p.oscillator_1.duty : 0.5f;
  o.o2_duty = (w2 == GROOVE_WAVE_PULSE_WIDTH) ? p.oscillator_2.duty : 0.5f;
  o.o1_duty64 = duty_to_u64((double)o.o1_duty);
  o.o2_duty64 = duty_to_u64((double)o.o2_duty);
  o.lfo_inc = turns_to_inc(p.lfo_frequency / sr);
  o.lfo_depth = p.lfo_depth;
  { // LFO recurrence constants and the WF_LFO_SMOOTH promise (dsp_core.h, welsh_frame)
    const double D = 6.28318530717958647692 * ((double)(int64_t)o.lfo_inc * 5.42101086242752217004e-20);
    const double h = sin(0.5 * D);
    o.lfo_rk = 2.0 * h * h;
    o.lfo_rs = sin(D);
    o.lfo_a = (double)p.lfo_depth * 0.693147180559945309417;
    const uint32_t wl = p.lfo_waveform & 15u, r = p.lfo_routing & 15u;
    const bool tri = wl == GROOVE_WAVE_TRIANGLE || wl == GROOVE_WAVE_TRIANGLE_SINE;
    const bool saw = wl == GROOVE_WAVE_SAWTOOTH, sq = wl == GROOVE_WAVE_SQUARE || wl == GROOVE_WAVE_PULSE_WIDTH;
    // largest per-frame change of the LFO value BETWEEN its edges: |D| for a sine, 4 |inc| (in turns) for a triangle, 2 |inc| for a
    // sawtooth, none for a square or a pulse (the frame of an edge is evaluated exactly: welsh_frame_front; round 6 — these three
    // waveforms ran in the exact-f64 kind before).  A noise LFO has no smooth stretch and stays exact.
    const double turns = fabs((double)(int64_t)o.lfo_inc * 5.42101086242752217004e-20);
    const double dl = wl == GROOVE_WAVE_SINE ? fabs(D) : (tri ? 4.0 * turns : (saw ? 2.0 * turns : 0.0));
    const uint32_t rb = lfo_routing_bits(r);
    if ((wl == GROOVE_WAVE_SINE || tri || saw || sq) &&
        ((rb & WF_LFO_PW) || ((rb & WF_LFO_PITCH) && fabs(o.lfo_a) * dl <= 1.5e-3)))
      o.flags |= WF_LFO_SMOOTH;
  }
  o.amp = derive_env(p.amp_envelope, sr);
  o.fil = derive_env(p.filter_envelope, sr);
  o.fc = derive_lp24_consts((double)p.filter_passband_ripple);
  // a retuned filter with c below 1/512 (ripple above ~3.8) takes its per-frame coefficients in the two-sided form (dsp_core.h WF_COEF_WIDE):
  // the one-sided form the fast kinds carry is measured 4e-7 off the oracle at ripple 3.5, 9e-7 at 4.0, 4e-6 at 5, 7e-5 at 7.1
  // (tools/reference_patches_emul.py: the reference's penny-whistle patch)
  if ((o.flags & (WF_RETUNE_ENV | WF_LFO_CUTOFF)) && o.fc.c0 < 1.0f / 512.0f) o.flags |= WF_COEF_WIDE;
  o.ripple = p.filter_passband_ripple;
  o.cutoff_hz = p.filter_cutoff_hz;
  o.cutoff_start = p.filter_cutoff_start;
  o.cutoff_end = p.filter_cutoff_end;
  pan_gains(p.dca_gain, p.dca_pan, o.gl, o.gr);
  cold.tune1 = p.oscillator_1.tune; cold.tune2 = p.oscillator_2.tune;
  cold.fixed1 = p.oscillator_1.fixed_hz; cold.fixed2 = p.oscillator_2.fixed_hz;
  return o;
}
// WF_FILTER_F32 (dsp_core.h "fp32 recurrence"): may this patch's 24 dB filter run in fp32?  MEASURED, not estimated: the fp32
// recurrence loses accuracy where a pole pair approaches the unit circle — at z = +1 (low cutoffs) and at z = -1 (cutoffs near
// SR/2) — by an amount that depends on the ripple too, and no closed form predicted the emulated voices' errors to better than
// 30x (docs/DSP_SPEC.md).  So the host runs the two recurrences side by side — the device's own coefficient formulas, a sawtooth
// at 110 Hz and at 1,760 Hz, 2,048 frames — at the lowest, the highest and the middle cutoff the patch can reach (static cutoff;
// envelope sweep start .. start + (1 - start) end; LFO sweep start (1 -+ depth)), and promises fp32 only if the worst RMS
// difference is <= 2e-6 of full scale.  The figure is NOT a bound on the voice's own error: low static cutoffs just under it play
// 2x off it (240 - 320 Hz at ripple 1.61: measured 1.7e-6 - 2.3e-6, voices 1.9e-6 - 4.8e-6), which is the margin the 2e-6 keeps
// to the path's 1e-5; at 5e-6 (tried at the end of round 5: three more benchmark patches, 2 % of the million-voice step) a static
// 40 Hz slipped through at 4.3e-6 and its voices were off by 4e-5.  The 32 benchmark patches: see docs/DSP_SPEC.md section 11;
// tests/test_emul_numerics.py holds the bars.  ~0.6 ms per distinct patch.
inline double welsh_filter_f32_error(const WelshParams& o, double sr) {
  const RenderConsts rc = render_consts(sr);
  float lo_fc, hi_fc;
  const auto fc_of_pct = [](float pct) { return 25.0f * powf(800.0f, fminf(fmaxf(pct, 0.0f), 1.0f)); };
  if (o.flags & WF_RETUNE_ENV) {
    lo_fc = fc_of_pct(o.cutoff_start); hi_fc = fc_of_pct(o.cutoff_start + (1.0f - o.cutoff_start) * o.cutoff_end);
  } else if (o.flags & WF_LFO_CUTOFF) {
    lo_fc = fc_of_pct(o.cutoff_start * (1.0f - fabsf(o.lfo_depth))); hi_fc = fc_of_pct(o.cutoff_start * (1.0f + fabsf(o.lfo_depth)));
  } else {
    lo_fc = hi_fc = o.cutoff_hz;
  }
  if (lo_fc > hi_fc) { const float t = lo_fc; lo_fc = hi_fc; hi_fc = t; }
  lo_fc = fminf(fmaxf(lo_fc, 1.0f), rc.fc_max); hi_fc = fminf(fmaxf(hi_fc, 1.0f), rc.fc_max);
  const float cut[3] = {lo_fc, sqrtf(lo_fc * hi_fc), hi_fc};
  // Round-off noise is not a smooth function of the cutoff (a static 40 Hz at ripple 1.61 reads 4e-6 where 38 and 42 Hz read 3e-5,
  // and the voice itself is off by 4e-5): every cutoff is measured with two neighbours, 6 % either side, and the worst counts.
  const float near[3] = {1.0f, 0.94f, 1.06f};
  const double pitch[2] = {110.0, 1760.0};
  constexpr int kWarm = 512, kFrames = 2048;
  double worst = 0.0;
  for (int ci = 0; ci < (lo_fc == hi_fc ? 1 : 3); ++ci) {
    for (float nb : near) {
      const float fc = fminf(fmaxf(cut[ci] * nb, 1.0f), rc.fc_max);
      const Lp24CoefD cd = lp24_coefd_from_fc(o.fc, fc, rc.pi_over_sr, rc.fc_max);
      const Lp24CoefF cf = lp24_coeff_from_fc(o.fc, fc, rc.pi_over_sr, rc.fc_max);
      for (double f0 : pitch) {
        Lp24StateD sd{0.0, 0.0, 0.0, 0.0};
        Lp24StateF sf{0.0f, 0.0f, 0.0f, 0.0f};
        double acc = 0.0, ph = 0.0;
        const double dph = f0 / sr;
        for (int i = 0; i < kWarm + kFrames; ++i) {
          const float x = (float)(ph - 0.5); // a sawtooth of amplitude 0.5: an oscillator mix's level
          ph += dph; if (ph >= 1.0) ph -= 1.0;
          const double yd = lp24_step(sd, cd, (double)x);
          const double yf = (double)lp24_step_f32(sf, cf, x);
          if (i >= kWarm) acc += (yf - yd) * (yf - yd);
        }
        const double rms = sqrt(acc / kFrames);
        if (!(rms <= worst)) worst = rms; // (NaN counts as failure)
      }
    }
  }
  return worst;
}
constexpr double kFilterF32MaxError = 2e-6; // (5e-6 / 6e-6 / 1e-5 were measured and rejected: profiles/r05_f32_threshold_ab.log, docs/DSP_SPEC.md section 11)
inline bool welsh_filter_f32_ok(const WelshParams& o, double sr) {
  if (o.flags & (WF_LFO_RESO | WF_COEF_WIDE)) return false; // the ripple moves every frame / the two-sided coefficient form: the exact-f64 kind
  const double e = welsh_filter_f32_error(o, sr);
  return e == e && e <= kFilterF32MaxError;
}
inline WelshState initial_welsh_state() {
  WelshState s{};
  osc_reset(s.o1); osc_reset(s.o2); osc_reset(s.lfo);
  env_init(s.amp); env_init(s.fil);
  s.vflags = VF_FIRST;
  return s;
}

inline FmParams derive_fm(const groove_fm_params& p, double sr) {
  FmParams o{};
  o.depth_beta = (double)p.depth * (double)p.beta;
  o.cenv = derive_env(p.carrier_envelope, sr);
  o.menv = derive_env(p.modulator_envelope, sr);
  pan_gains(p.dca_gain, p.dca_pan, o.gl, o.gr);
  return o;
}
inline FmState initial_fm_state() {
  FmState s{};
  osc_reset(s.carrier); osc_reset(s.modulator);
  env_init(s.cenv); env_init(s.menv);
  s.vflags = VF_FIRST;
  return s;
}

// note_to_frequency(key): 12-TET, A4 = 440 Hz (settings/src/patches.rs:8,96).
GROOVE_HD double note_to_frequency(uint32_t key) { return 440.0 * exp2(((double)key - 69.0) / 12.0); }

// PlaysNotes::note_on / note_off bodies, shared by the device event kernels.
GROOVE_HD void welsh_note(const WelshParams& p, WelshState& s, double tune1, double tune2,
                          double fixed1, double fixed2, double sr, uint32_t key, bool on) {
  if (on) {
    const double f = note_to_frequency(key);
    s.o1_inc = turns_to_inc((fixed1 > 0.0 ? fixed1 : f * tune1) / sr);
    s.o2_inc = turns_to_inc((fixed2 > 0.0 ? fixed2 : f * tune2) / sr);
    env_trigger_attack(s.amp, p.amp);
    env_trigger_attack(s.fil, p.fil);
  } else {
    env_trigger_release(s.amp, p.amp);
    env_trigger_release(s.fil, p.fil);
  }
}
GROOVE_HD void fm_note(const FmParams& p, FmState& s, double ratio, double sr, uint32_t key, bool on) {
  if (on) {
    const double f = note_to_frequency(key);
    s.c_inc = turns_to_inc(f / sr);
    s.m_inc = turns_to_inc(f * ratio / sr);
    env_trigger_attack(s.cenv, p.cenv);
    env_trigger_attack(s.menv, p.menv);
  } else {
    env_trigger_release(s.cenv, p.cenv);
    env_trigger_release(s.menv, p.menv);
  }
}
GROOVE_HD void sampler_note(const SamplerParams& p, SamplerState& s, uint32_t key, bool on) {
  if (on) {
    s.playing = 1; s.idx = 0;
    const double step = p.root_hz > 0.0 ? note_to_frequency(key) / p.root_hz : 1.0;
    s.step = (uint64_t)(step * 17592186044416.0); // 2^44
  } else if (!p.one_shot) {
    s.playing = 0;
  }
}

// f64 coefficient sets for the effect kernels (host; identical to the oracle's math).
inline void rbj_lowpass_h(double f0, double q, double fs, double* c5) {
  const double w0 = 2.0 * 3.14159265358979323846 * f0 / fs, cw = cos(w0), sw = sin(w0);
  const double alpha = sw / (2.0 * q), a0 = 1.0 + alpha;
  c5[0] = (1.0 - cw) / 2.0 / a0; c5[1] = (1.0 - cw) / a0; c5[2] = (1.0 - cw) / 2.0 / a0;
  c5[3] = -2.0 * cw / a0; c5[4] = (1.0 - alpha) / a0;
}
inline void rbj_highpass_h(double f0, double q, double fs, double* c5) {
  const double w0 = 2.0 * 3.14159265358979323846 * f0 / fs, cw = cos(w0), sw = sin(w0);
  const double alpha = sw / (2.0 * q), a0 = 1.0 + alpha;
  c5[0] = (1.0 + cw) / 2.0 / a0; c5[1] = -(1.0 + cw) / a0; c5[2] = (1.0 + cw) / 2.0 / a0;
  c5[3] = -2.0 * cw / a0; c5[4] = (1.0 - alpha) / a0;
}
// The remaining cookbook modes (doc/Audio-EQ-Cookbook.txt:113-198); parameter conventions: DSP_SPEC §4.
inline double bw_octaves_h(double f0, double bw_hz) {
  const double lo = f0 - 0.5 * bw_hz, hi = f0 + 0.5 * bw_hz;
  if (!(lo > 0.0) || hi / lo > 256.0) return 8.0;
  return log2(hi / lo);
}
inline void rbj_norm_h(double b0, double b1, double b2, double a0, double a1, double a2, double* c5) {
  c5[0] = b0 / a0; c5[1] = b1 / a0; c5[2] = b2 / a0; c5[3] = a1 / a0; c5[4] = a2 / a0;
}
// Coefficients of any BiQuad 12 dB effect kind; false when `kind` is not one.
inline bool rbj_for_kind_h(uint32_t kind, const groove_fx_params& p, double fs, double* c5) {
  const double pi = 3.14159265358979323846;
  const double f0 = p.cutoff_hz, w0 = 2.0 * pi * f0 / fs, cw = cos(w0), sw = sin(w0);
  switch (kind) {
    case GROOVE_FX_BIQUAD_LP12: rbj_lowpass_h(f0, p.q, fs, c5); return true;
    case GROOVE_FX_BIQUAD_HP12: rbj_highpass_h(f0, p.q, fs, c5); return true;
    case GROOVE_FX_BIQUAD_BP12: {
      const double al = sw * sinh(log(2.0) / 2.0 * bw_octaves_h(f0, p.bandwidth_hz) * w0 / sw);
      rbj_norm_h(al, 0.0, -al, 1.0 + al, -2.0 * cw, 1.0 - al, c5); return true;
    }
    case GROOVE_FX_BIQUAD_BS12: {
      const double al = sw * sinh(log(2.0) / 2.0 * bw_octaves_h(f0, p.bandwidth_hz) * w0 / sw);
      rbj_norm_h(1.0, -2.0 * cw, 1.0, 1.0 + al, -2.0 * cw, 1.0 - al, c5); return true;
    }
    case GROOVE_FX_BIQUAD_AP12: {
      const double al = sw / (2.0 * p.q);
      rbj_norm_h(1.0 - al, -2.0 * cw, 1.0 + al, 1.0 + al, -2.0 * cw, 1.0 - al, c5); return true;
    }
    case GROOVE_FX_BIQUAD_PEAK12: {
      const double A = pow(10.0, p.db_gain / 40.0), al = sw / (2.0 * 0.70710678118654752440);
      rbj_norm_h(1.0 + al * A, -2.0 * cw, 1.0 - al * A, 1.0 + al / A, -2.0 * cw, 1.0 - al / A, c5); return true;
    }
    case GROOVE_FX_BIQUAD_LSHELF12: {
      const double A = pow(10.0, p.db_gain / 40.0), t = 2.0 * sqrt(A) * (sw / 2.0 * sqrt(2.0));
      rbj_norm_h(A * ((A + 1) - (A - 1) * cw + t), 2 * A * ((A - 1) - (A + 1) * cw), A * ((A + 1) - (A - 1) * cw - t),
                 (A + 1) + (A - 1) * cw + t, -2 * ((A - 1) + (A + 1) * cw), (A + 1) + (A - 1) * cw - t, c5);
      return true;
    }
    case GROOVE_FX_BIQUAD_HSHELF12: {
      const double A = pow(10.0, p.db_gain / 40.0), t = 2.0 * sqrt(A) * (sw / 2.0 * sqrt(2.0));
      rbj_norm_h(A * ((A + 1) + (A - 1) * cw + t), -2 * A * ((A - 1) + (A + 1) * cw), A * ((A + 1) + (A - 1) * cw - t),
                 (A + 1) - (A - 1) * cw + t, 2 * ((A - 1) - (A + 1) * cw), (A + 1) - (A - 1) * cw - t, c5);
      return true;
    }
    default: return false;
  }
}
// out6 = b0,a1,a2 (section 1), b0,a1,a2 (section 2); y = b0 x + 2 b0 x1 + b0 x2 + a1 y1 + a2 y2
inline void lp24_coeffs_h(double fc, double ripple, double fs, double* out6) {
  if (fc > 0.49 * fs) fc = 0.49 * fs;
  if (fc < 1.0) fc = 1.0;
  const double k = tan(3.14159265358979323846 * fc / fs);
  double sg = sinh(ripple), cg = cosh(ripple);
  cg *= cg;
  const double c0 = 1.0 / (cg - 0.85355339059327376220), c1 = k * c0 * sg * 1.84775906502257351226;
  const double c2 = 1.0 / (cg - 0.14644660940672623780), c3 = k * c2 * sg * 0.76536686473017954346;
  const double K = k * k;
  const double a0 = 1.0 / (c1 + K + c0), a3 = 1.0 / (c3 + K + c2);
  out6[0] = a0 * K; out6[1] = 2.0 * (c0 - K) * a0; out6[2] = (c1 - K - c0) * a0;
  out6[3] = a3 * K; out6[4] = 2.0 * (c2 - K) * a3; out6[5] = (c3 - K - c2) * a3;
}
inline uint32_t delay_frames_h(double seconds, double sr) {
  const double n = floor(seconds * sr + 0.5);
  if (!(n >= 1.0)) return 1u;                    // zero, negative, NaN
  return n > 1073741824.0 ? 1073741824u : (uint32_t)n; // the ring allocation fails long before this
}
inline double decay_gain_h(double delay_s, double decay_s) {
  return decay_s > 0.0 ? pow(0.001, delay_s / decay_s) : 0.0;
}
static const double kCombDelaysH[4] = {0.0297, 0.0371, 0.0411, 0.0437};
static const double kAllpassDelaysH[2] = {0.005, 0.0017};
static const double kAllpassDecaysH[2] = {0.09683, 0.03292};
inline double percent_to_frequency_h(double p) { return 25.0 * pow(800.0, p); }

} // namespace groove
