// oracle_dsp.hpp — CPU restatement (f64, scalar, one frame at a time) of the
// reference's per-voice DSP and mix bus.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (groove_amd/, the C ABI
// in include/groove_hip.h) may include, link or call this code.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
// or as the timed CPU baseline — never as the thing shipped.
//
// PARITY STATUS.  The reference snapshot does not contain the DSP arithmetic: it
// lives in the un-vendored, un-pinned path dependency `ensnare` / `groove-core`
// (/root/reference/Cargo.toml:26-30, no Cargo.lock) and no Rust
// toolchain exists here.  What IS pinned, and checked by tests/test_oracle_*.py:
//   * mix-bus / Gain / chain / fan-in sums       orchestration/src/orchestrator.rs:1444-1668
//   * render frame counts                        orchestrator.rs:1689-1737, 1822-1827, 1903-1908
//   * semis_and_cents / octaves                  settings/src/patches.rs:754-796
//   * MMA concave/convex transforms              orchestration/src/util.rs:4-21, 286-318
//   * RBJ biquad coefficients + Direct Form 1    doc/Audio-EQ-Cookbook.txt:38-39, 76-111
//   * bilinear-transformed 2-section 24 dB LPF   doc/filters004.txt:70-116, 307-408
//     (that C text is compiled from where it lies into oracle/_ref, see oracle/Makefile)
//   * 16-bit WAV quantisation                    orchestration/src/helpers.rs:74-97
// Everything else (oscillator waveforms and noise, envelope shape, the Chebyshev
// 24 dB formula, denormalize_q, frequency_to_percent, Dca pan law, FM, sampler
// stepping, bitcrusher, chorus, delay, reverb) is "PARITY UNPINNED": this file and
// docs/DSP_SPEC.md ARE the definition, following SURVEY.md Appendix A.
//
// Each class mirrors the reference trait surface: Ticks::tick, Generates::value,
// TransformsAudio::transform_channel, Configurable::update_sample_rate,
// PlaysNotes::note_on/note_off (entities/src/instruments/metronome.rs:23-60 shows
// the call order: tick() then value()).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "../include/groove_types.h"

namespace oracle {

constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------- helpers (a17)
// settings/src/patches.rs:8,96 — MIDI key → Hz, 12-TET, A4 (key 69) = 440 Hz.
inline double note_to_frequency(int key) { return 440.0 * std::exp2((key - 69) / 12.0); }
// settings/src/patches.rs:255-258.
inline double semis_and_cents(int semitones, double cents) {
  return std::pow(2.0, (semitones * 100.0 + cents) / 1200.0);
}
// settings/src/patches.rs:251-253.
inline double octaves(int n) { return semis_and_cents(n * 12, 0.0); }
// FrequencyHz::frequency_to_percent / percent_to_frequency (patches.rs:150-152); Appendix A.4.
inline double percent_to_frequency(double p) { return 25.0 * std::pow(800.0, p); }
inline double frequency_to_percent(double f) { return std::log(f / 25.0) / std::log(800.0); }
// BiQuadFilter::denormalize_q (patches.rs:148); Appendix A.4.
inline double denormalize_q(double n) { return n * n * 10.0 + 0.707; }
// orchestration/src/util.rs:4-11.
inline double mma_concave(double x) {
  if (x > 1.0 - std::pow(10.0, -12.0 / 5.0)) return 1.0;
  return -(5.0 / 12.0) * std::log10(1.0 - x);
}
// orchestration/src/util.rs:14-21.
inline double mma_convex(double x) {
  if (x < std::pow(10.0, -12.0 / 5.0)) return 0.0;
  return 1.0 + (5.0 / 12.0) * std::log10(x);
}
// orchestration/src/helpers.rs:79-91: (x * 32767.0) as i16 — Rust `as` truncates
// toward zero and saturates; NaN → 0.
inline int16_t wav_quantise(double x) {
  double v = x * 32767.0;
  if (std::isnan(v)) return 0;
  if (v >= 32767.0) return 32767;
  if (v <= -32768.0) return -32768;
  return (int16_t)v; // C++ truncation toward zero == Rust `as` in range
}
inline double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// ---------------------------------------------------------------- Oscillator (a1)
// Constructed at settings/src/patches.rs:104-107, 260-266; ticked at
// entities/src/instruments/metronome.rs:48-59.  DSP_SPEC §2.
struct Oscillator {
  uint32_t waveform = GROOVE_WAVE_SINE;
  double duty = 0.5;
  double frequency = 440.0;  // set_frequency
  double tune = 1.0;         // set_frequency_tune (Ratio)
  double fixed_hz = 0.0;     // set_fixed_frequency; > 0 overrides frequency*tune
  double fm = 0.0;           // set_frequency_modulation (exponent, 2^fm)
  double lfm = 0.0;          // set_linear_frequency_modulation
  double duty_eff = 0.5;     // duty after LFO pulse-width routing
  double sample_rate = GROOVE_DEFAULT_SAMPLE_RATE;
  // state
  double pos = 0.0;
  bool first = true;
  bool sync_pending = false;
  bool should_sync_ = false;
  uint32_t x1 = 0x70f4f854u, x2 = 0xe1e9f0a7u; // musicdsp "fast white noise" seeds
  double noise_value = 0.0;

  void configure(const groove_oscillator_params& p) {
    waveform = p.waveform;
    duty = (p.waveform == GROOVE_WAVE_PULSE_WIDTH) ? (double)p.duty : 0.5;
    duty_eff = duty;
    tune = p.tune;
    fixed_hz = p.fixed_hz;
  }
  void update_sample_rate(double sr) { sample_rate = sr; }
  void sync() { sync_pending = true; }
  bool should_sync() const { return should_sync_; }
  double adjusted_frequency() const {
    double base = fixed_hz > 0.0 ? fixed_hz : frequency * tune;
    return base * (std::exp2(fm) + lfm);
  }
  // Ticks::tick(1)
  void tick() {
    double delta = adjusted_frequency() / sample_rate;
    if (sync_pending) {
      pos = 0.0;
      sync_pending = false;
      first = false;
      should_sync_ = false;
    } else if (first) {
      first = false; // first tick after reset emits position 0
      should_sync_ = false;
    } else {
      pos += delta;
      double w = std::floor(pos);
      should_sync_ = (w != 0.0);
      pos -= w;
    }
    if (waveform == GROOVE_WAVE_NOISE) {
      x1 ^= x2;
      noise_value = (double)(int32_t)x2 * (1.0 / 2147483648.0);
      x2 += x1;
    }
  }
  // Generates::value()
  double value() const {
    const double p = pos;
    switch (waveform) {
      case GROOVE_WAVE_SINE: return std::sin(2.0 * kPi * p);
      case GROOVE_WAVE_SQUARE:
      case GROOVE_WAVE_PULSE_WIDTH: return p < duty_eff ? 1.0 : -1.0;
      case GROOVE_WAVE_TRIANGLE: return 4.0 * std::fabs(p - std::floor(p + 0.5)) - 1.0;
      case GROOVE_WAVE_SAWTOOTH: return 2.0 * (p - std::floor(p + 0.5));
      case GROOVE_WAVE_TRIANGLE_SINE:
        return 4.0 * std::fabs(p - std::floor(p + 0.75) + 0.25) - 1.0;
      case GROOVE_WAVE_NOISE: return noise_value;
      case GROOVE_WAVE_DEBUG_MAX: return 1.0;
      case GROOVE_WAVE_DEBUG_MIN: return -1.0;
      default: return 0.0; // None, DebugZero
    }
  }
};

// ---------------------------------------------------------------- Envelope (a2)
// EnvelopeParams at settings/src/patches.rs:133-138, 154-159.  DSP_SPEC §3.
struct Envelope {
  enum State : uint32_t { IDLE = 0, ATTACK = 1, DECAY = 2, SUSTAIN = 3, RELEASE = 4 };
  double attack = 0, decay = 0, sustain = 1, release = 0; // seconds / Normal
  double sample_rate = GROOVE_DEFAULT_SAMPLE_RATE;
  uint32_t state = IDLE;
  double A = 0, B = 0; // stage start / target level
  double len = 0;      // stage length, frames (full-scale time × distance)
  uint32_t N = 0, n = 0;
  double value_ = 0.0;

  void configure(const groove_envelope_params& p) {
    attack = p.attack; decay = p.decay; sustain = clamp01(p.sustain); release = p.release;
  }
  void update_sample_rate(double sr) { sample_rate = sr; }
  // N = ceil(len (1 - 2^-16)): the stage's frame count, taken a hair below `len` so that a length which IS a whole number of frames in
  // exact arithmetic — 0.3 s x 44,100 x 0.6 = 7,938, what round patch values give — stays that number whichever side of it binary
  // rounding puts `len` (f64 here, fp32 on the device: docs/DSP_SPEC.md section 3, round 5).
  static uint32_t frames(double len) {
    if (!(len > 0.0)) return 0;
    double c = std::ceil(len * (1.0 - 1.0 / 65536.0));
    return c > 4.0e9 ? 4000000000u : (uint32_t)c;
  }
  void enter(uint32_t st, double from) {
    state = st;
    n = 0;
    if (st == ATTACK) { A = from; B = 1.0; len = attack * sample_rate * (1.0 - A); }
    else if (st == DECAY) { A = from; B = sustain; len = decay * sample_rate * (A - B); }
    else if (st == RELEASE) { A = from; B = 0.0; len = release * sample_rate * A; }
    N = frames(len);
  }
  void trigger_attack() { enter(ATTACK, value_); }
  void trigger_release() { if (state != IDLE) enter(RELEASE, value_); }
  bool is_idle() const { return state == IDLE; }
  void tick() {
    for (;;) {
      if (state == IDLE) { value_ = 0.0; return; }
      if (state == SUSTAIN) { value_ = sustain; return; }
      if (n >= N) {
        if (state == ATTACK) enter(DECAY, 1.0);
        else if (state == DECAY) state = SUSTAIN;
        else state = IDLE;
        continue;
      }
      double t = (double)n / len;
      value_ = A + (B - A) * (2.0 * t - t * t);
      ++n;
      return;
    }
  }
  double value() const { return value_; }
};

// ---------------------------------------------------------------- BiQuad 12 dB (a3)
// doc/Audio-EQ-Cookbook.txt:76-111 (coefficients), :38-39 Eq 4 (Direct Form 1).
struct BiquadCoeffs { double b0, b1, b2, a1, a2; }; // already divided by a0
inline BiquadCoeffs rbj_lowpass(double f0, double q, double fs) {
  double w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw / (2.0 * q);
  double a0 = 1.0 + alpha;
  return {(1.0 - cw) / 2.0 / a0, (1.0 - cw) / a0, (1.0 - cw) / 2.0 / a0, -2.0 * cw / a0,
          (1.0 - alpha) / a0};
}
inline BiquadCoeffs rbj_highpass(double f0, double q, double fs) {
  double w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw / (2.0 * q);
  double a0 = 1.0 + alpha;
  return {(1.0 + cw) / 2.0 / a0, -(1.0 + cw) / a0, (1.0 + cw) / 2.0 / a0, -2.0 * cw / a0,
          (1.0 - alpha) / a0};
}
// The remaining cookbook modes (doc/Audio-EQ-Cookbook.txt:113-198).  Parameter conventions of the
// project files (projects/demos/effects/filter-*-12db_*.json): band-pass / band-stop carry a
// `bandwidth` in Hz, converted to the cookbook's BW in octaves between the -3 dB frequencies,
// BW = log2((f0 + bw/2) / (f0 - bw/2)) (capped at 8 octaves when bw >= 2 f0); peaking and shelves
// carry `db-gain`; peaking uses Q = 1/sqrt(2), shelves use slope S = 1.  [conventions unpinned,
// coefficient formulas pinned]
inline double bw_octaves(double f0, double bw_hz) {
  const double lo = f0 - 0.5 * bw_hz, hi = f0 + 0.5 * bw_hz;
  if (!(lo > 0.0) || hi / lo > 256.0) return 8.0;
  return std::log2(hi / lo);
}
inline BiquadCoeffs rbj_normalise(double b0, double b1, double b2, double a0, double a1, double a2) {
  return {b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
}
inline BiquadCoeffs rbj_bandpass(double f0, double bw_hz, double fs) { // constant 0 dB peak gain
  double w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw * std::sinh(std::log(2.0) / 2.0 * bw_octaves(f0, bw_hz) * w0 / sw);
  return rbj_normalise(alpha, 0.0, -alpha, 1.0 + alpha, -2.0 * cw, 1.0 - alpha);
}
inline BiquadCoeffs rbj_bandstop(double f0, double bw_hz, double fs) {
  double w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw * std::sinh(std::log(2.0) / 2.0 * bw_octaves(f0, bw_hz) * w0 / sw);
  return rbj_normalise(1.0, -2.0 * cw, 1.0, 1.0 + alpha, -2.0 * cw, 1.0 - alpha);
}
inline BiquadCoeffs rbj_allpass(double f0, double q, double fs) {
  double w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0), alpha = sw / (2.0 * q);
  return rbj_normalise(1.0 - alpha, -2.0 * cw, 1.0 + alpha, 1.0 + alpha, -2.0 * cw, 1.0 - alpha);
}
inline BiquadCoeffs rbj_peaking(double f0, double db_gain, double fs) {
  double A = std::pow(10.0, db_gain / 40.0), w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw / (2.0 * 0.70710678118654752440);
  return rbj_normalise(1.0 + alpha * A, -2.0 * cw, 1.0 - alpha * A, 1.0 + alpha / A, -2.0 * cw, 1.0 - alpha / A);
}
inline BiquadCoeffs rbj_lowshelf(double f0, double db_gain, double fs) {
  double A = std::pow(10.0, db_gain / 40.0), w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw / 2.0 * std::sqrt(2.0), t = 2.0 * std::sqrt(A) * alpha; // S = 1
  return rbj_normalise(A * ((A + 1) - (A - 1) * cw + t), 2 * A * ((A - 1) - (A + 1) * cw), A * ((A + 1) - (A - 1) * cw - t),
                       (A + 1) + (A - 1) * cw + t, -2 * ((A - 1) + (A + 1) * cw), (A + 1) + (A - 1) * cw - t);
}
inline BiquadCoeffs rbj_highshelf(double f0, double db_gain, double fs) {
  double A = std::pow(10.0, db_gain / 40.0), w0 = 2.0 * kPi * f0 / fs, cw = std::cos(w0), sw = std::sin(w0);
  double alpha = sw / 2.0 * std::sqrt(2.0), t = 2.0 * std::sqrt(A) * alpha;
  return rbj_normalise(A * ((A + 1) + (A - 1) * cw + t), -2 * A * ((A - 1) + (A + 1) * cw), A * ((A + 1) + (A - 1) * cw - t),
                       (A + 1) - (A - 1) * cw + t, 2 * ((A - 1) - (A + 1) * cw), (A + 1) - (A - 1) * cw - t);
}
inline bool rbj_for_kind(uint32_t kind, const groove_fx_params& p, double sr, BiquadCoeffs& out) {
  switch (kind) {
    case GROOVE_FX_BIQUAD_LP12: out = rbj_lowpass(p.cutoff_hz, p.q, sr); return true;
    case GROOVE_FX_BIQUAD_HP12: out = rbj_highpass(p.cutoff_hz, p.q, sr); return true;
    case GROOVE_FX_BIQUAD_BP12: out = rbj_bandpass(p.cutoff_hz, p.bandwidth_hz, sr); return true;
    case GROOVE_FX_BIQUAD_BS12: out = rbj_bandstop(p.cutoff_hz, p.bandwidth_hz, sr); return true;
    case GROOVE_FX_BIQUAD_AP12: out = rbj_allpass(p.cutoff_hz, p.q, sr); return true;
    case GROOVE_FX_BIQUAD_PEAK12: out = rbj_peaking(p.cutoff_hz, p.db_gain, sr); return true;
    case GROOVE_FX_BIQUAD_LSHELF12: out = rbj_lowshelf(p.cutoff_hz, p.db_gain, sr); return true;
    case GROOVE_FX_BIQUAD_HSHELF12: out = rbj_highshelf(p.cutoff_hz, p.db_gain, sr); return true;
    default: return false;
  }
}
struct BiquadDF1 { // one channel
  double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
  double step(const BiquadCoeffs& c, double x) {
    double y = c.b0 * x + c.b1 * x1 + c.b2 * x2 - c.a1 * y1 - c.a2 * y2;
    x2 = x1; x1 = x; y2 = y1; y1 = y;
    return y;
  }
};

// ---------------------------------------------------------------- 24 dB low-pass (a4)
// BiQuadFilterLowPass24db{cutoff, passband_ripple}: settings/src/patches.rs:146-149,
// settings/src/effects.rs:41,83-85.  Formula: SURVEY Appendix A.4 (two cascaded
// Chebyshev-style sections; the s-plane constants are the 4th-order pole pairs printed
// in doc/filters004.txt:141-155, 231-244).  Each section is the bilinear transform
// (k = tan(pi fc/fs)) of  H(s) = 1 / (c s^2 + d s + 1)  with
//   section 1: c = c0 = 1/(cosh^2 r - 0.853553..), d = c0 sinh r 1.847759..
//   section 2: c = c2 = 1/(cosh^2 r - 0.146446..), d = c2 sinh r 0.765366..
// which tests/test_oracle_ref.py checks against doc/filters004.txt's szxform().
struct Lp24Coeffs { double b0[2], a1[2], a2[2]; }; // b1 = 2 b0, b2 = b0
inline Lp24Coeffs lp24_coeffs(double fc, double ripple, double fs) {
  if (fc > 0.49 * fs) fc = 0.49 * fs;
  if (fc < 1.0) fc = 1.0;
  double k = std::tan(kPi * fc / fs);
  double sg = std::sinh(ripple), cg = std::cosh(ripple);
  cg *= cg;
  double c0 = 1.0 / (cg - 0.85355339059327376220);
  double c1 = k * c0 * sg * 1.84775906502257351226;
  double c2 = 1.0 / (cg - 0.14644660940672623780);
  double c3 = k * c2 * sg * 0.76536686473017954346;
  double K = k * k;
  Lp24Coeffs c;
  double a0 = 1.0 / (c1 + K + c0);
  c.a1[0] = 2.0 * (c0 - K) * a0; c.a2[0] = (c1 - K - c0) * a0; c.b0[0] = a0 * K;
  double a3 = 1.0 / (c3 + K + c2);
  c.a1[1] = 2.0 * (c2 - K) * a3; c.a2[1] = (c3 - K - c2) * a3; c.b0[1] = a3 * K;
  return c;
}
struct Lp24State { // one channel, transposed direct form II, two sections
  double s[4] = {0, 0, 0, 0};
  double step(const Lp24Coeffs& c, double x) {
    double y1 = c.b0[0] * x + s[0];
    s[0] = 2.0 * c.b0[0] * x + c.a1[0] * y1 + s[1];
    s[1] = c.b0[0] * x + c.a2[0] * y1;
    double y2 = c.b0[1] * y1 + s[2];
    s[2] = 2.0 * c.b0[1] * y1 + c.a1[1] * y2 + s[3];
    s[3] = c.b0[1] * y1 + c.a2[1] * y2;
    return y2;
  }
};

// ---------------------------------------------------------------- Dca (a13)
// DcaParams{gain, pan}: settings/src/patches.rs:160-168.  Pan law: Appendix A.7.
inline void dca(double x, double gain, double pan, double& L, double& R) {
  double l = 1.0 - 0.25 * (pan + 1.0) * (pan + 1.0);
  double r = 1.0 - (0.5 * pan - 0.5) * (0.5 * pan - 0.5);
  L = x * gain * l;
  R = x * gain * r;
}

// ---------------------------------------------------------------- WelshVoice (a5)
// WelshVoiceParams: settings/src/patches.rs:110-164; frame order Appendix A.6.
struct WelshVoice {
  groove_welsh_params p{};
  Oscillator o1, o2, lfo;
  Envelope amp, fil;
  Lp24State filt;
  Lp24Coeffs coeffs{};
  double sample_rate = GROOVE_DEFAULT_SAMPLE_RATE;
  double L = 0, R = 0;

  void configure(const groove_welsh_params& params, double sr) {
    p = params;
    sample_rate = sr;
    o1 = Oscillator(); o2 = Oscillator(); lfo = Oscillator();
    o1.configure(p.oscillator_1); o2.configure(p.oscillator_2);
    lfo.waveform = p.lfo_waveform; lfo.frequency = p.lfo_frequency;
    amp.configure(p.amp_envelope); fil.configure(p.filter_envelope);
    update_sample_rate(sr);
  }
  void update_sample_rate(double sr) {
    sample_rate = sr;
    o1.update_sample_rate(sr); o2.update_sample_rate(sr); lfo.update_sample_rate(sr);
    amp.update_sample_rate(sr); fil.update_sample_rate(sr);
    coeffs = lp24_coeffs(p.filter_cutoff_hz, p.filter_passband_ripple, sr);
  }
  void note_on(int key, int /*velocity*/) {
    double f = note_to_frequency(key);
    o1.frequency = f; o2.frequency = f;
    amp.trigger_attack(); fil.trigger_attack();
  }
  void note_off(int /*velocity*/) { amp.trigger_release(); fil.trigger_release(); }
  bool is_playing() const { return !amp.is_idle(); }
  void tick() {
    amp.tick(); fil.tick();
    if (amp.is_idle()) { L = R = 0.0; return; }
    lfo.tick();
    const double l = lfo.value();
    const double depth = p.lfo_depth;
    const uint32_t r = p.lfo_routing;
    // docs/DSP_SPEC.md §6: which oscillators an edge-moving routing reaches
    if (r == GROOVE_LFO_PITCH) { o1.fm = l * depth; o2.fm = l * depth; }
    if (r == GROOVE_LFO_PITCH_OSC2) o2.fm = l * depth;
    if (r == GROOVE_LFO_PULSE_WIDTH || r == GROOVE_LFO_PW_OSC1) o1.duty_eff = clamp01(o1.duty * (1.0 + l * depth));
    if (r == GROOVE_LFO_PULSE_WIDTH || r == GROOVE_LFO_PW_OSC2) o2.duty_eff = clamp01(o2.duty * (1.0 + l * depth));
    o1.tick();
    if (p.oscillator_2_sync && o1.should_sync()) o2.sync();
    o2.tick();
    const double mix = p.oscillator_mix;
    double s = o1.value() * mix + o2.value() * (1.0 - mix);
    bool retune = false;
    double fc = p.filter_cutoff_hz;
    if (p.filter_cutoff_end != 0.0f) {
      fc = percent_to_frequency(clamp01(p.filter_cutoff_start + (1.0 - p.filter_cutoff_start) * p.filter_cutoff_end * fil.value()));
      retune = true;
    } else if (r == GROOVE_LFO_FILTER_CUTOFF || r == GROOVE_LFO_CUTOFF_AMP) {
      fc = percent_to_frequency(clamp01(p.filter_cutoff_start * (1.0 + l * depth)));
      retune = true;
    }
    double ripple = p.filter_passband_ripple;
    if (r == GROOVE_LFO_RESONANCE) { ripple *= (1.0 + l * depth); retune = true; }
    if (retune) coeffs = lp24_coeffs(fc, ripple, sample_rate);
    double y = filt.step(coeffs, s);
    double a = amp.value();
    if (r == GROOVE_LFO_AMPLITUDE || r == GROOVE_LFO_CUTOFF_AMP) a *= (1.0 + l * depth);
    dca(y * a, p.dca_gain, p.dca_pan, L, R);
  }
};

// ---------------------------------------------------------------- FmVoice (a6)
// FmSynthParams: settings/src/patches.rs:691-715; Appendix A.11.
struct FmVoice {
  groove_fm_params p{};
  Oscillator carrier, modulator;
  Envelope cenv, menv;
  double L = 0, R = 0;
  void configure(const groove_fm_params& params, double sr) {
    p = params;
    carrier = Oscillator(); modulator = Oscillator();
    cenv.configure(p.carrier_envelope); menv.configure(p.modulator_envelope);
    update_sample_rate(sr);
  }
  void update_sample_rate(double sr) {
    carrier.update_sample_rate(sr); modulator.update_sample_rate(sr);
    cenv.update_sample_rate(sr); menv.update_sample_rate(sr);
  }
  void note_on(int key, int) {
    double f = note_to_frequency(key);
    carrier.frequency = f; modulator.frequency = f * p.ratio;
    cenv.trigger_attack(); menv.trigger_attack();
  }
  void note_off(int) { cenv.trigger_release(); menv.trigger_release(); }
  void tick() {
    cenv.tick(); menv.tick();
    if (cenv.is_idle()) { L = R = 0.0; return; }
    modulator.tick();
    carrier.lfm = modulator.value() * menv.value() * (double)p.depth * (double)p.beta;
    carrier.tick();
    dca(carrier.value() * cenv.value(), p.dca_gain, p.dca_pan, L, R);
  }
};

// ---------------------------------------------------------------- SamplerVoice (a7)
// SamplerParams{filename, root}, Drumkit{name}: settings/src/instruments.rs:34-37, 81-88;
// pointer stepping without interpolation: README.md:82-85, Appendix A.10.
struct SamplerVoice {
  const float* pcm = nullptr; // buffer start inside the shared bank
  uint32_t length = 0;
  double root_hz = 0.0;
  bool one_shot = true;
  double gain = 1.0;
  bool playing = false;
  double idx = 0.0, step = 1.0;
  double L = 0, R = 0;
  void note_on(int key, int) {
    playing = true; idx = 0.0;
    step = root_hz > 0.0 ? note_to_frequency(key) / root_hz : 1.0;
  }
  void note_off(int) { if (!one_shot) playing = false; }
  void tick() {
    if (!playing) { L = R = 0.0; return; }
    double fi = std::floor(idx);
    if (fi >= (double)length) { playing = false; L = R = 0.0; return; }
    double s = (double)pcm[(size_t)fi] * gain;
    L = R = s; // mono duplicated to both channels
    idx += step;
  }
};

// ---------------------------------------------------------------- effects (a8-a12)
// TransformsAudio::transform_channel(channel, sample); state is per channel.
inline double gain_fx(double x, double ceiling) { return x * ceiling; } // a8, effects.rs:25,73

// a9 Bitcrusher, Appendix A.8.  The quantise is integer arithmetic on the 16-bit scale
// and is defined on the f32 value of the input so that GPU and CPU agree bit for bit.
inline float bitcrush_f32(float x, uint32_t bits) {
  float ax = std::fabs(x) * 32767.0f;
  if (!(ax < 2147483648.0f)) ax = 2147483520.0f;
  uint32_t q = (uint32_t)ax; // truncation
  q = (q >> bits) << bits;
  float y = (float)q * (1.0f / 32767.0f);
  return std::copysign(y, x);
}

struct Ring { // DelayLine, Appendix A.9
  std::vector<double> buf;
  uint32_t idx = 0;
  void resize(uint32_t n) { buf.assign(n < 1 ? 1 : n, 0.0); idx = 0; }
  uint32_t size() const { return (uint32_t)buf.size(); }
  double peek(uint32_t delay_back) const { // delay_back in 1..size: sample written that many pushes ago
    uint32_t n = size();
    return buf[(idx + n - (delay_back % n)) % n];
  }
  double oldest() const { return buf[idx]; }
  void push(double v) { buf[idx] = v; idx = (idx + 1) % size(); }
};
inline uint32_t delay_frames(double seconds, double sr) {
  double n = std::floor(seconds * sr + 0.5);
  return n < 1.0 ? 1u : (uint32_t)n;
}

struct DelayFx { // a11 Delay{seconds}: effects.rs:35,107-109
  Ring ring;
  void configure(double seconds, double sr) { ring.resize(delay_frames(seconds, sr)); }
  double step(double x) { double y = ring.oldest(); ring.push(x); return y; }
};

struct ChorusFx { // a10 Chorus{voices, delay_seconds}: effects.rs:31,113-115
  Ring ring; uint32_t voices = 1, spacing = 0;
  void configure(uint32_t v, double seconds, double sr) {
    voices = v < 1 ? 1 : v;
    ring.resize(delay_frames(seconds, sr));
    spacing = ring.size() / voices;
  }
  double step(double x) {
    // tap k reads the sample pushed (N - k*spacing) frames ago; k = 0 is the full line.
    double sum = 0.0;
    uint32_t n = ring.size();
    for (uint32_t k = 0; k < voices; ++k) sum += ring.peek(n - k * spacing);
    ring.push(x);
    return sum;
  }
};

// a12 Reverb{attenuation, seconds}: effects.rs:37,110-112; Schroeder topology, A.9.
constexpr double kCombDelays[4] = {0.0297, 0.0371, 0.0411, 0.0437};
constexpr double kAllpassDelays[2] = {0.005, 0.0017};
constexpr double kAllpassDecays[2] = {0.09683, 0.03292};
inline double decay_gain(double delay_s, double decay_s) {
  return decay_s > 0.0 ? std::pow(0.001, delay_s / decay_s) : 0.0;
}
struct ReverbFx {
  Ring comb[4], ap[2];
  double g_comb[4], g_ap[2];
  double attenuation = 1.0;
  void configure(double att, double seconds, double sr) {
    attenuation = att;
    for (int i = 0; i < 4; ++i) {
      comb[i].resize(delay_frames(kCombDelays[i], sr));
      g_comb[i] = decay_gain(kCombDelays[i], seconds);
    }
    for (int i = 0; i < 2; ++i) {
      ap[i].resize(delay_frames(kAllpassDelays[i], sr));
      g_ap[i] = decay_gain(kAllpassDelays[i], kAllpassDecays[i]);
    }
  }
  double step(double x) {
    double in = x * attenuation;
    double sum = 0.0;
    for (int i = 0; i < 4; ++i) { // recirculating comb: out = g*oldest; push(in + out)
      double out = g_comb[i] * comb[i].oldest();
      comb[i].push(in + out);
      sum += out;
    }
    for (int i = 0; i < 2; ++i) { // Schroeder all-pass
      double d = ap[i].oldest();
      double v = sum + g_ap[i] * d;
      ap[i].push(v);
      sum = d - g_ap[i] * v;
    }
    return sum;
  }
};

inline double limiter_fx(double x, double mn, double mx) { // A.8
  double a = std::fabs(x);
  a = a < mn ? mn : (a > mx ? mx : a);
  return std::copysign(a, x);
}
inline double compressor_fx(double x, double threshold, double ratio) { // A.8
  double a = std::fabs(x);
  if (a > threshold) a = threshold + (a - threshold) * ratio;
  return std::copysign(a, x);
}

// One effect instance for one lane (stereo = two channel states).
struct Effect {
  uint32_t kind = GROOVE_FX_MIXER;
  groove_fx_params p{};
  double sr = GROOVE_DEFAULT_SAMPLE_RATE;
  BiquadCoeffs bq{}; BiquadDF1 df1[2];
  Lp24Coeffs l24{}; Lp24State l24s[2];
  DelayFx delay[2]; ChorusFx chorus[2]; ReverbFx reverb[2];
  void configure(uint32_t k, const groove_fx_params& params, double sample_rate) {
    kind = k; p = params; sr = sample_rate;
    retune();
    for (int c = 0; c < 2; ++c) {
      if (kind == GROOVE_FX_DELAY) delay[c].configure(p.delay_seconds, sr);
      if (kind == GROOVE_FX_CHORUS) chorus[c].configure(p.voices, p.delay_seconds, sr);
      if (kind == GROOVE_FX_REVERB) reverb[c].configure(p.attenuation, p.reverb_seconds, sr);
    }
  }
  void retune() {
    rbj_for_kind(kind, p, sr, bq);
    if (kind == GROOVE_FX_BIQUAD_LP24) l24 = lp24_coeffs(p.cutoff_hz, p.passband_ripple, sr);
    for (int c = 0; c < 2; ++c) reverb[c].attenuation = p.attenuation;
  }
  double wet_of(int ch, double x) {
    switch (kind) {
      case GROOVE_FX_GAIN: return gain_fx(x, p.ceiling);
      case GROOVE_FX_BITCRUSHER: return (double)bitcrush_f32((float)x, p.bits);
      case GROOVE_FX_BIQUAD_LP12:
      case GROOVE_FX_BIQUAD_HP12:
      case GROOVE_FX_BIQUAD_BP12:
      case GROOVE_FX_BIQUAD_BS12:
      case GROOVE_FX_BIQUAD_AP12:
      case GROOVE_FX_BIQUAD_PEAK12:
      case GROOVE_FX_BIQUAD_LSHELF12:
      case GROOVE_FX_BIQUAD_HSHELF12: return df1[ch].step(bq, x);
      case GROOVE_FX_BIQUAD_LP24: return l24s[ch].step(l24, x);
      case GROOVE_FX_CHORUS: return chorus[ch].step(x);
      case GROOVE_FX_DELAY: return delay[ch].step(x);
      case GROOVE_FX_REVERB: return reverb[ch].step(x);
      case GROOVE_FX_LIMITER: return limiter_fx(x, p.limit_min, p.limit_max);
      case GROOVE_FX_COMPRESSOR: return compressor_fx(x, p.limit_min, p.limit_max);
      default: return x; // Mixer: identity
    }
  }
  double transform_channel(int ch, double x) {
    double w = wet_of(ch, x);
    if (p.wet >= 1.0f) return w;
    return x * (1.0 - (double)p.wet) + w * (double)p.wet;
  }
};

} // namespace oracle
