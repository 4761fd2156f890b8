// oracle_api.cpp — extern "C" surface of the CPU oracle (see oracle_dsp.hpp header
// for the TEST-INFRASTRUCTURE-ONLY rule and the parity status).
//
// Buffers are f64, planar, frame-major: block[ch][frame][voice] (same layout as the
// GPU blocks) and bus[frame][2].
#include "oracle_dsp.hpp"
#include <cstring>
#include <memory>
#include <thread>
#include <functional>

using namespace oracle;

namespace {

struct Bank { // an instrument: n homogeneous voices (Synthesizer + voice store)
  enum Kind { WELSH, FM, SAMPLER } kind;
  double sr;
  std::vector<WelshVoice> welsh;
  std::vector<FmVoice> fm;
  std::vector<SamplerVoice> sampler;
  std::vector<float> pcm; // sampler bank copy
  uint32_t n() const {
    return (uint32_t)(kind == WELSH ? welsh.size() : kind == FM ? fm.size() : sampler.size());
  }
  void note(const groove_note_event& e) {
    uint32_t lo = e.voice, hi = e.voice + 1;
    if (e.voice == GROOVE_ALL_VOICES) { lo = 0; hi = n(); }
    for (uint32_t v = lo; v < hi && v < n(); ++v) {
      if (kind == WELSH) { if (e.on) welsh[v].note_on(e.key, e.velocity); else welsh[v].note_off(e.velocity); }
      else if (kind == FM) { if (e.on) fm[v].note_on(e.key, e.velocity); else fm[v].note_off(e.velocity); }
      else { if (e.on) sampler[v].note_on(e.key, e.velocity); else sampler[v].note_off(e.velocity); }
    }
  }
  inline void tick_voice(uint32_t v, double& L, double& R) {
    if (kind == WELSH) { welsh[v].tick(); L = welsh[v].L; R = welsh[v].R; }
    else if (kind == FM) { fm[v].tick(); L = fm[v].L; R = fm[v].R; }
    else { sampler[v].tick(); L = sampler[v].L; R = sampler[v].R; }
  }
};

struct FxBank {
  std::vector<Effect> lanes;
};

// Orchestrator graph (orchestration/src/orchestrator.rs:367-470, 1198, 1326-1347).
struct Node {
  enum Type { SOURCE_CONST, INSTRUMENT, EFFECT, TOY_NEGATE } type;
  double level = 0.0;             // ToyAudioSource{level}
  std::unique_ptr<Bank> bank;     // instrument: sum of its voices
  Effect fx;                      // effect
  std::vector<int> sources;       // audio_sink_uid_to_source_uids
};
// ControlTrip (entities/src/controllers/control_trip.rs:7-26: the step shapes; :99-142 add_path: a step
// covers [cursor, cursor + path_multiplier) beats and interpolates start → end by its function;
// :184-254 work: the value is sent to the target whenever it changed).  The GPU path applies automation
// once per block (Orchestrator::tick, orchestrator.rs:856-859: handle_work once, then gather_audio over
// the whole buffer), with the value the trip has at the block's first frame — restated here the same way.
struct TripStep { int kind; double start, end, beats; };
struct Trip {
  int target; uint32_t index; double start_beat;
  std::vector<TripStep> steps;
  double last_sent = -1.0;
};
struct Graph {
  double sr;
  std::vector<Node> nodes; // uid = index; uid 0 = main mixer
  double bpm = 128.0;
  uint64_t clock_frames = 0;
  std::vector<Trip> trips;
};
// Value of one step at t in [0, 1]: Flat, Slope (linear), Logarithmic (fast first: the MMA convex
// transform of the ramp), Exponential (slow first: the MMA concave transform); docs/DSP_SPEC.md §7.
double trip_step_value(int kind, double start, double end, double t) {
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  switch (kind) {
    case 0: return start;
    case 1: return start + (end - start) * t;
    case 2: return start + (end - start) * mma_convex(t);
    case 3: return start + (end - start) * mma_concave(t);
    default: return start;
  }
}
// Controllable::control_set_param_by_index for an effect: ControlValue 0..1 → the parameter's own unit
// (include/groove_types.h groove_control_index).
void effect_set_control(Effect& fx, uint32_t index, double v) {
  v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
  switch (index) {
    case GROOVE_CTL_FX_CEILING: fx.p.ceiling = (float)v; break;
    case GROOVE_CTL_FX_BITS: fx.p.bits = (uint32_t)(v * 16.0); break;
    case GROOVE_CTL_FX_CUTOFF: fx.p.cutoff_hz = (float)percent_to_frequency(v); break;
    case GROOVE_CTL_FX_Q: fx.p.q = (float)denormalize_q(v); break;
    case GROOVE_CTL_FX_PASSBAND_RIPPLE: fx.p.passband_ripple = (float)denormalize_q(v); break;
    case GROOVE_CTL_FX_ATTENUATION: fx.p.attenuation = (float)v; break;
    case GROOVE_CTL_FX_WET: fx.p.wet = (float)v; break;
    case GROOVE_CTL_FX_THRESHOLD: fx.p.limit_min = (float)v; break;
    default: return;
  }
  fx.retune();
}

} // namespace

extern "C" {

// ---------------------------------------------------------------- scalar helpers
double oracle_note_to_frequency(int key) { return note_to_frequency(key); }
double oracle_semis_and_cents(int semis, double cents) { return semis_and_cents(semis, cents); }
double oracle_octaves(int n) { return octaves(n); }
double oracle_percent_to_frequency(double p) { return percent_to_frequency(p); }
double oracle_frequency_to_percent(double f) { return frequency_to_percent(f); }
double oracle_denormalize_q(double n) { return denormalize_q(n); }
double oracle_mma_concave(double x) { return mma_concave(x); }
double oracle_mma_convex(double x) { return mma_convex(x); }
int oracle_wav_quantise(double x) { return wav_quantise(x); }
float oracle_bitcrush_f32(float x, uint32_t bits) { return bitcrush_f32(x, bits); }
void oracle_dca(double x, double gain, double pan, double* lr) { dca(x, gain, pan, lr[0], lr[1]); }

// out[5] = b0,b1,b2,a1,a2 (all divided by a0)
void oracle_rbj_lowpass(double f0, double q, double fs, double* out) {
  BiquadCoeffs c = rbj_lowpass(f0, q, fs);
  out[0] = c.b0; out[1] = c.b1; out[2] = c.b2; out[3] = c.a1; out[4] = c.a2;
}
void oracle_rbj_highpass(double f0, double q, double fs, double* out) {
  BiquadCoeffs c = rbj_highpass(f0, q, fs);
  out[0] = c.b0; out[1] = c.b1; out[2] = c.b2; out[3] = c.a1; out[4] = c.a2;
}
// any BiQuad 12 dB kind: out[5] = b0,b1,b2,a1,a2; returns 0 when `kind` is not a biquad
int oracle_rbj_for_kind(uint32_t kind, const groove_fx_params* p, double fs, double* out) {
  BiquadCoeffs c;
  if (!rbj_for_kind(kind, *p, fs, c)) return 0;
  out[0] = c.b0; out[1] = c.b1; out[2] = c.b2; out[3] = c.a1; out[4] = c.a2;
  return 1;
}
// out[6] = b0,a1,a2 of section 1 then section 2
void oracle_lp24_coeffs(double fc, double ripple, double fs, double* out) {
  Lp24Coeffs c = lp24_coeffs(fc, ripple, fs);
  out[0] = c.b0[0]; out[1] = c.a1[0]; out[2] = c.a2[0];
  out[3] = c.b0[1]; out[4] = c.a1[1]; out[5] = c.a2[1];
}
void oracle_lp24_run(double fc, double ripple, double fs, const double* x, double* y, uint32_t n) {
  Lp24Coeffs c = lp24_coeffs(fc, ripple, fs);
  Lp24State s;
  for (uint32_t i = 0; i < n; ++i) y[i] = s.step(c, x[i]);
}
void oracle_biquad_df1_run(const double* coeffs5, const double* x, double* y, uint32_t n) {
  BiquadCoeffs c{coeffs5[0], coeffs5[1], coeffs5[2], coeffs5[3], coeffs5[4]};
  BiquadDF1 s;
  for (uint32_t i = 0; i < n; ++i) y[i] = s.step(c, x[i]);
}

// Oscillator alone: n frames of value() after tick(); fm_per_frame may be NULL.
void oracle_oscillator_run(const groove_oscillator_params* p, double frequency, double sr,
                           const double* fm_per_frame, double* out, uint32_t n,
                           uint32_t* noise_state_out /* x1,x2 or NULL */) {
  Oscillator o;
  o.configure(*p);
  o.frequency = frequency;
  o.update_sample_rate(sr);
  for (uint32_t i = 0; i < n; ++i) {
    if (fm_per_frame) o.fm = fm_per_frame[i];
    o.tick();
    out[i] = o.value();
  }
  if (noise_state_out) { noise_state_out[0] = o.x1; noise_state_out[1] = o.x2; }
}

// Envelope alone: note-on at frame 0, note-off at frame `off_frame` (or never if >= n).
void oracle_envelope_run(const groove_envelope_params* p, double sr, uint32_t off_frame,
                         double* out, uint32_t n) {
  Envelope e;
  e.configure(*p);
  e.update_sample_rate(sr);
  for (uint32_t i = 0; i < n; ++i) {
    if (i == 0) e.trigger_attack();
    if (i == off_frame) e.trigger_release();
    e.tick();
    out[i] = e.value();
  }
}

// Orchestrator::run / run_performance frame counts (orchestrator.rs:788-846).
// A Timer controller finishes after `beats` beats; handle_work (:631-708) completes
// frames only up to the end of the performance, so the final tick() returns
// ticks_completed = total % buffer.  `run` keeps the partial block (:795),
// `run_performance` drops it (:827-836).
uint64_t oracle_performance_total_frames(double beats, double bpm, double sr) {
  return (uint64_t)std::ceil(beats * 60.0 / bpm * sr);
}
uint64_t oracle_run_frames(double beats, double bpm, double sr, uint32_t /*buffer*/) {
  return oracle_performance_total_frames(beats, bpm, sr);
}
uint64_t oracle_run_performance_frames(double beats, double bpm, double sr, uint32_t buffer) {
  uint64_t total = oracle_performance_total_frames(beats, bpm, sr);
  return total - total % buffer;
}

// ---------------------------------------------------------------- instrument banks
void* oracle_welsh_create(const groove_welsh_params* p, uint32_t n, uint32_t sr) {
  Bank* b = new Bank();
  b->kind = Bank::WELSH; b->sr = sr;
  b->welsh.resize(n);
  for (uint32_t i = 0; i < n; ++i) b->welsh[i].configure(p[i], sr);
  return b;
}
void* oracle_fm_create(const groove_fm_params* p, uint32_t n, uint32_t sr) {
  Bank* b = new Bank();
  b->kind = Bank::FM; b->sr = sr;
  b->fm.resize(n);
  for (uint32_t i = 0; i < n; ++i) b->fm[i].configure(p[i], sr);
  return b;
}
void* oracle_sampler_create(const float* bank_pcm, uint64_t bank_frames, const groove_sample_desc* d,
                            uint32_t n_samples, const groove_sampler_params* p, uint32_t n, uint32_t sr) {
  Bank* b = new Bank();
  b->kind = Bank::SAMPLER; b->sr = sr;
  b->pcm.assign(bank_pcm, bank_pcm + bank_frames);
  b->sampler.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    const groove_sample_desc& sd = d[p[i].sample_index < n_samples ? p[i].sample_index : 0];
    b->sampler[i].pcm = b->pcm.data() + sd.offset;
    b->sampler[i].length = sd.length;
    b->sampler[i].root_hz = sd.root_hz;
    b->sampler[i].one_shot = p[i].one_shot != 0;
    b->sampler[i].gain = p[i].gain;
  }
  return b;
}
void oracle_bank_destroy(void* h) { delete (Bank*)h; }
void oracle_bank_note_events(void* h, const groove_note_event* ev, uint32_t n_ev) {
  Bank* b = (Bank*)h;
  for (uint32_t i = 0; i < n_ev; ++i) b->note(ev[i]);
}
// Generates::generate_batch_values: out[2][frames][n] f64.
// Controllable on an instrument (include/groove_hip.h groove_bank_set_param: Welsh banks; ControlValue 0..1 -> dca gain, pan as a
// BipolarNormal, the static filter cutoff through percent_to_frequency), applied between blocks with the voices' state untouched —
// what `#[derive(Control)]` generates for the reference's entities (proc-macros/src/control.rs:171-183).  Returns 0, or -1 for an
// instrument or index that has no control.
int oracle_bank_set_param(void* h, uint32_t voice, uint32_t control_index, double value01) {
  Bank* b = (Bank*)h;
  if (b->kind != Bank::WELSH) return -1;
  const double v01 = value01 < 0.0 ? 0.0 : (value01 > 1.0 ? 1.0 : value01);
  uint32_t lo = voice, hi = voice + 1;
  if (voice == GROOVE_ALL_VOICES) { lo = 0; hi = b->n(); }
  for (uint32_t v = lo; v < hi && v < b->n(); ++v) {
    WelshVoice& w = b->welsh[v];
    switch (control_index) {
      case GROOVE_CTL_WELSH_DCA_GAIN: w.p.dca_gain = (float)v01; break;
      case GROOVE_CTL_WELSH_DCA_PAN: w.p.dca_pan = (float)(v01 * 2.0 - 1.0); break;
      case GROOVE_CTL_WELSH_CUTOFF:
        w.p.filter_cutoff_hz = (float)percent_to_frequency(v01);
        w.coeffs = lp24_coeffs(w.p.filter_cutoff_hz, w.p.filter_passband_ripple, w.sample_rate);
        break;
      default: return -1;
    }
  }
  return 0;
}
void oracle_bank_render(void* h, uint32_t frames, double* out) {
  Bank* b = (Bank*)h;
  const uint32_t n = b->n();
  for (uint32_t f = 0; f < frames; ++f)
    for (uint32_t v = 0; v < n; ++v) {
      double L, R;
      b->tick_voice(v, L, R);
      out[(size_t)f * n + v] = L;
      out[((size_t)frames + f) * n + v] = R;
    }
}
// Render and sum straight into bus[frames][2] (+=), voice range [v0, v1).  Frame-major
// like the reference (orchestrator.rs:367-410: per frame, tick every leaf, sum).
void oracle_bank_render_bus_range(void* h, uint32_t frames, double* bus, uint32_t v0, uint32_t v1) {
  Bank* b = (Bank*)h;
  for (uint32_t f = 0; f < frames; ++f) {
    double sl = 0.0, sr_ = 0.0;
    for (uint32_t v = v0; v < v1; ++v) {
      double L, R;
      b->tick_voice(v, L, R);
      sl += L; sr_ += R;
    }
    bus[2 * f] += sl; bus[2 * f + 1] += sr_;
  }
}
void oracle_bank_render_bus(void* h, uint32_t frames, double* bus) {
  Bank* b = (Bank*)h;
  oracle_bank_render_bus_range(h, frames, bus, 0, b->n());
}
// CPU-baseline mode B: voices sharded over `threads` host threads,
// per-thread partial buses summed at the end.
void oracle_bank_render_bus_mt(void* h, uint32_t frames, double* bus, uint32_t threads) {
  Bank* b = (Bank*)h;
  const uint32_t n = b->n();
  if (threads < 1) threads = 1;
  std::vector<std::vector<double>> part(threads, std::vector<double>((size_t)frames * 2, 0.0));
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < threads; ++t) {
    uint32_t v0 = (uint32_t)((uint64_t)n * t / threads), v1 = (uint32_t)((uint64_t)n * (t + 1) / threads);
    th.emplace_back([=, &part]() { oracle_bank_render_bus_range(h, frames, part[t].data(), v0, v1); });
  }
  for (auto& x : th) x.join();
  for (uint32_t t = 0; t < threads; ++t)
    for (size_t i = 0; i < (size_t)frames * 2; ++i) bus[i] += part[t][i];
}
// The same with PERSISTENT workers (round 6; bench.py's all-cores baseline): every thread owns its voices for `blocks` consecutive
// blocks of `frames` frames between one spawn and one join, adding into a bus of its own; bus[blocks][frames][2] (+)= the sum.
// (oracle_bank_render_bus_mt spawns and joins once per block: with 64 voices per thread that times the threads, not the voices.)
void oracle_bank_render_bus_blocks_mt(void* h, uint32_t frames, uint32_t blocks, double* bus, uint32_t threads) {
  Bank* b = (Bank*)h;
  const uint32_t n = b->n();
  if (threads < 1) threads = 1;
  if (threads > n) threads = n ? n : 1;
  const size_t per = (size_t)frames * 2, total = per * blocks;
  std::vector<std::vector<double>> part(threads, std::vector<double>(total, 0.0));
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < threads; ++t) {
    uint32_t v0 = (uint32_t)((uint64_t)n * t / threads), v1 = (uint32_t)((uint64_t)n * (t + 1) / threads);
    th.emplace_back([=, &part]() { for (uint32_t k = 0; k < blocks; ++k) oracle_bank_render_bus_range(h, frames, part[t].data() + per * k, v0, v1); });
  }
  for (auto& x : th) x.join();
  for (uint32_t t = 0; t < threads; ++t)
    for (size_t i = 0; i < total; ++i) bus[i] += part[t][i];
}
unsigned oracle_hardware_concurrency() { return std::thread::hardware_concurrency(); }

// ---------------------------------------------------------------- effect banks
void* oracle_fx_create(uint32_t kind, const groove_fx_params* p, uint32_t n, uint32_t sr) {
  FxBank* fx = new FxBank();
  fx->lanes.resize(n);
  for (uint32_t i = 0; i < n; ++i) fx->lanes[i].configure(kind, p[i], sr);
  return fx;
}
void oracle_fx_destroy(void* h) { delete (FxBank*)h; }
void oracle_fx_set_params(void* h, const groove_fx_params* p, uint32_t n) {
  FxBank* fx = (FxBank*)h;
  for (uint32_t i = 0; i < n && i < fx->lanes.size(); ++i) { fx->lanes[i].p = p[i]; fx->lanes[i].retune(); }
}
// TransformsAudio over a block, in place: inout[2][frames][n] f64.
void oracle_fx_process(void* h, double* inout, uint32_t frames) {
  FxBank* fx = (FxBank*)h;
  const uint32_t n = (uint32_t)fx->lanes.size();
  for (uint32_t f = 0; f < frames; ++f)
    for (uint32_t v = 0; v < n; ++v)
      for (int ch = 0; ch < 2; ++ch) {
        size_t i = ((size_t)ch * frames + f) * n + v;
        inout[i] = fx->lanes[v].transform_channel(ch, inout[i]);
      }
}
// Mix bus over materialised blocks: bus[f][ch] (+)= sum_v block[ch][f][v].
void oracle_mix(const double* block, uint32_t n, uint32_t frames, double* bus, int accumulate) {
  for (uint32_t f = 0; f < frames; ++f)
    for (int ch = 0; ch < 2; ++ch) {
      double s = 0.0;
      const double* row = block + ((size_t)ch * frames + f) * n;
      for (uint32_t v = 0; v < n; ++v) s += row[v];
      if (accumulate) bus[2 * f + ch] += s; else bus[2 * f + ch] = s;
    }
}

// ---------------------------------------------------------------- orchestrator graph
void* oracle_graph_create(uint32_t sr) {
  Graph* g = new Graph();
  g->sr = sr;
  g->nodes.emplace_back();
  g->nodes[0].type = Node::EFFECT; // main mixer: identity effect (orchestrator.rs:543-546)
  groove_fx_params p{}; p.wet = 1.0f;
  g->nodes[0].fx.configure(GROOVE_FX_MIXER, p, sr);
  return g;
}
void oracle_graph_destroy(void* h) { delete (Graph*)h; }
int oracle_graph_add_source_const(void* h, double level) {
  Graph* g = (Graph*)h;
  g->nodes.emplace_back();
  g->nodes.back().type = Node::SOURCE_CONST; g->nodes.back().level = level;
  return (int)g->nodes.size() - 1;
}
int oracle_graph_add_toy_effect(void* h) { // ToyEffect negates
  Graph* g = (Graph*)h;
  g->nodes.emplace_back();
  g->nodes.back().type = Node::TOY_NEGATE;
  return (int)g->nodes.size() - 1;
}
int oracle_graph_add_effect(void* h, uint32_t kind, const groove_fx_params* p) {
  Graph* g = (Graph*)h;
  g->nodes.emplace_back();
  g->nodes.back().type = Node::EFFECT;
  g->nodes.back().fx.configure(kind, *p, g->sr);
  return (int)g->nodes.size() - 1;
}
int oracle_graph_add_instrument(void* h, void* bank /* ownership moves to the graph */) {
  Graph* g = (Graph*)h;
  g->nodes.emplace_back();
  g->nodes.back().type = Node::INSTRUMENT;
  g->nodes.back().bank.reset((Bank*)bank);
  return (int)g->nodes.size() - 1;
}
int oracle_graph_patch(void* h, int source_uid, int sink_uid) { // Orchestrator::patch, :263-304
  Graph* g = (Graph*)h;
  if (source_uid < 0 || sink_uid < 0 || source_uid >= (int)g->nodes.size() || sink_uid >= (int)g->nodes.size()) return -1;
  if (source_uid == sink_uid) return -1;
  Node& sink = g->nodes[sink_uid];
  if (sink.type != Node::EFFECT && sink.type != Node::TOY_NEGATE) return -1; // only effects have inputs
  sink.sources.push_back(source_uid);
  return 0;
}
void oracle_graph_unpatch_all(void* h) {
  Graph* g = (Graph*)h;
  for (auto& n : g->nodes) n.sources.clear();
}
void oracle_graph_note_events(void* h, int uid, const groove_note_event* ev, uint32_t n_ev) {
  Graph* g = (Graph*)h;
  if (uid > 0 && uid < (int)g->nodes.size() && g->nodes[uid].bank)
    for (uint32_t i = 0; i < n_ev; ++i) g->nodes[uid].bank->note(ev[i]);
}
// ---- automation + the block loop (Orchestrator::tick, orchestrator.rs:856-877)
void oracle_graph_set_bpm(void* h, double bpm) { ((Graph*)h)->bpm = bpm; }
void oracle_graph_skip_to_start(void* h) {
  Graph* g = (Graph*)h;
  g->clock_frames = 0;
  for (auto& t : g->trips) t.last_sent = -1.0;
}
int oracle_graph_add_control_trip(void* h, int target_uid, uint32_t control_index, double start_beat) {
  Graph* g = (Graph*)h;
  if (target_uid <= 0 || target_uid >= (int)g->nodes.size() || g->nodes[target_uid].type != Node::EFFECT) return -1;
  g->trips.push_back(Trip{target_uid, control_index, start_beat, {}, -1.0});
  return (int)g->trips.size() - 1;
}
int oracle_graph_trip_add_step(void* h, int trip, int kind, double start, double end, double beats) {
  Graph* g = (Graph*)h;
  if (trip < 0 || trip >= (int)g->trips.size() || !(beats > 0.0)) return -1;
  g->trips[trip].steps.push_back(TripStep{kind, start, end, beats});
  return 0;
}
double oracle_control_step_value(int kind, double start, double end, double t) { return trip_step_value(kind, start, end, t); }
void oracle_graph_gather(void* h, uint32_t frames, double* bus);
// One tick(): the trips' values at the block's first frame (MusicalTime has 65,536 units per beat,
// src/mini/transport.rs:157-176; the clock converts frames to whole units), then gather_audio over the
// block, then the clock advances.
void oracle_graph_tick(void* h, uint32_t frames, double* bus) {
  Graph* g = (Graph*)h;
  const uint64_t units = (uint64_t)((double)g->clock_frames * g->bpm / 60.0 / g->sr * 65536.0);
  const double now = (double)units / 65536.0;
  for (auto& t : g->trips) {
    double b = t.start_beat;
    for (const TripStep& s : t.steps) {
      if (now >= b && now < b + s.beats) {
        const double v = trip_step_value(s.kind, s.start, s.end, (now - b) / s.beats);
        if (v != t.last_sent) { t.last_sent = v; effect_set_control(g->nodes[t.target].fx, t.index, v); }
        break;
      }
      b += s.beats;
    }
  }
  oracle_graph_gather(h, frames, bus);
  g->clock_frames += frames;
}

// gather_audio, orchestrator.rs:367-470: per frame, explicit-stack post-order DFS from
// the main mixer; a leaf instrument is ticked once and its value added to the running
// sum; an effect sums ALL its sources, transforms that sum once, and adds the result to
// the sum accumulated before it was visited.
void oracle_graph_gather(void* h, uint32_t frames, double* bus) {
  Graph* g = (Graph*)h;
  struct Entry { bool collect; int uid; double al, ar; };
  std::vector<Entry> stack;
  for (uint32_t f = 0; f < frames; ++f) {
    double sl = 0.0, sr_ = 0.0;
    stack.clear();
    stack.push_back({false, 0, 0, 0});
    while (!stack.empty()) {
      Entry e = stack.back();
      stack.pop_back();
      Node& nd = g->nodes[e.uid];
      if (!e.collect) {
        if (nd.type == Node::SOURCE_CONST) { sl += nd.level; sr_ += nd.level; }
        else if (nd.type == Node::INSTRUMENT) {
          Bank* b = nd.bank.get();
          for (uint32_t v = 0; v < b->n(); ++v) { double L, R; b->tick_voice(v, L, R); sl += L; sr_ += R; }
        } else {
          stack.push_back({true, e.uid, sl, sr_});
          sl = sr_ = 0.0;
          for (int s : nd.sources) stack.push_back({false, s, 0, 0});
        }
      } else {
        double tl, tr;
        if (nd.type == Node::TOY_NEGATE) { tl = -sl; tr = -sr_; }
        else { tl = nd.fx.transform_channel(0, sl); tr = nd.fx.transform_channel(1, sr_); }
        sl = e.al + tl; sr_ = e.ar + tr;
      }
    }
    bus[2 * f] = sl; bus[2 * f + 1] = sr_;
  }
}

} // extern "C"
