// project.cpp — see project.hpp.
#include "project.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace groove_host {

namespace {

std::string slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("couldn't read " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
double note_to_frequency(int key) { return 440.0 * std::pow(2.0, (key - 69) / 12.0); }
double semis_and_cents(int semis, double cents) { return std::pow(2.0, (semis * 100.0 + cents) / 1200.0); }
double frequency_to_percent(double f) { return std::log(f / 25.0) / std::log(800.0); }
double denormalize_q(double n) { return n * n * 10.0 + 0.707; }
double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// BeatValueSettings (settings/src/lib.rs:121-157): the enum discriminant is 4096 / beats.
double beat_value_beats(const std::string& name) {
  static const std::pair<const char*, double> table[] = {
      {"octuple", 128}, {"quadruple", 256}, {"double", 512}, {"whole", 1024}, {"half", 2048}, {"quarter", 4096},
      {"eighth", 8192}, {"sixteenth", 16384}, {"thirty-second", 32768}, {"sixty-fourth", 65536},
      {"one-hundred-twenty-eighth", 131072}, {"two-hundred-fifty-sixth", 262144}, {"five-hundred-twelfth", 524288}};
  for (auto& t : table) if (name == t.first) return 4096.0 / t.second;
  throw std::runtime_error("unknown note-value '" + name + "'");
}

// Waveform: "sine" | {"pulse-width": 0.25} | … (settings/src/patches.rs:173-189)
void parse_waveform(const json5::Value* v, uint32_t& waveform, float& duty) {
  waveform = GROOVE_WAVE_SINE; duty = 0.5f;
  if (!v) return;
  std::string name;
  if (v->is_string()) name = v->str;
  else if (v->is_object() && !v->obj.empty()) {
    name = v->obj[0].first;
    if (v->obj[0].second->is_number()) duty = (float)v->obj[0].second->num;
  }
  static const std::pair<const char*, uint32_t> table[] = {
      {"none", GROOVE_WAVE_NONE}, {"sine", GROOVE_WAVE_SINE}, {"square", GROOVE_WAVE_SQUARE},
      {"pulse-width", GROOVE_WAVE_PULSE_WIDTH}, {"triangle", GROOVE_WAVE_TRIANGLE}, {"sawtooth", GROOVE_WAVE_SAWTOOTH},
      {"noise", GROOVE_WAVE_NOISE}, {"debug-zero", GROOVE_WAVE_DEBUG_ZERO}, {"debug-max", GROOVE_WAVE_DEBUG_MAX},
      {"debug-min", GROOVE_WAVE_DEBUG_MIN}, {"triangle-sine", GROOVE_WAVE_TRIANGLE_SINE}};
  for (auto& t : table) if (name == t.first) { waveform = t.second; return; }
  throw std::runtime_error("unknown waveform '" + name + "'");
}
// OscillatorTune → Ratio (patches.rs:204-219); returns the note for OscillatorTune::Note, else -1.
int parse_tune(const json5::Value* v, double& ratio) {
  ratio = 1.0;
  if (!v) return -1;
  if (v->is_number()) { ratio = v->num; return -1; }
  if (!v->is_object() || v->obj.empty()) return -1;
  const std::string& k = v->obj[0].first;
  const json5::Value& x = *v->obj[0].second;
  if (k == "float") ratio = x.num;
  else if (k == "note") return (int)json5::to_int(x.num, -1, 127, -1);
  else if (k == "osc") ratio = semis_and_cents((int)json5::to_int(x.number_or("octave", 0), -16, 16) * 12 + (int)json5::to_int(x.number_or("semi", 0), -1200, 1200), x.number_or("cent", 0));
  return -1;
}
groove_envelope_params parse_envelope(const json5::Value* v) {
  groove_envelope_params e{0, 0, 1, 0};
  if (v) { e.attack = v->number_or("attack", 0); e.decay = v->number_or("decay", 0); e.sustain = v->number_or("sustain", 1); e.release = v->number_or("release", 0); }
  return e;
}

} // namespace

std::string patch_name_to_settings_name(const std::string& name) {
  // from_case(Camel).without_boundaries([DigitLower]).to_case(Kebab): split before an upper-case
  // letter that follows a lower-case letter or digit; already-kebab names pass through.
  std::string out;
  for (size_t i = 0; i < name.size(); ++i) {
    const char c = name[i];
    if (c >= 'A' && c <= 'Z') {
      if (i && ((name[i - 1] >= 'a' && name[i - 1] <= 'z') || (name[i - 1] >= '0' && name[i - 1] <= '9'))) out += '-';
      out += (char)(c - 'A' + 'a');
    } else if (c == ' ' || c == '_') out += '-';
    else out += c;
  }
  return out;
}

groove_welsh_params welsh_params_from_patch_json(const json5::Value& patch, std::vector<std::string>* warnings) {
  groove_welsh_params p{};
  const json5::Value* o1 = patch.get("oscillator-1");
  const json5::Value* o2 = patch.get("oscillator-2");
  parse_waveform(o1 ? o1->get("waveform") : nullptr, p.oscillator_1.waveform, p.oscillator_1.duty);
  parse_waveform(o2 ? o2->get("waveform") : nullptr, p.oscillator_2.waveform, p.oscillator_2.duty);
  parse_tune(o1 ? o1->get("tune") : nullptr, p.oscillator_1.tune);
  const int note2 = parse_tune(o2 ? o2->get("tune") : nullptr, p.oscillator_2.tune);
  const bool track2 = patch.bool_or("oscillator-2-track", true);
  if (p.oscillator_2.waveform != GROOVE_WAVE_NONE && !track2) {
    if (note2 < 0) throw std::runtime_error("patch has oscillator-2-track = false, so oscillator 2 needs a fixed pitch, but its tune is not given as a note");
    p.oscillator_2.fixed_hz = note_to_frequency(note2); // patches.rs:94-100
  }
  p.oscillator_2_sync = patch.bool_or("oscillator-2-sync", false) ? 1 : 0;
  // oscillator_mix (patches.rs:88-108, 123-132): the oscillator list also counts a noise source
  const double mix1 = o1 ? o1->number_or("mix-pct", 1.0) : 1.0, mix2 = o2 ? o2->number_or("mix-pct", 1.0) : 1.0;
  int n_osc = (p.oscillator_1.waveform != GROOVE_WAVE_NONE) + (p.oscillator_2.waveform != GROOVE_WAVE_NONE) +
              (patch.number_or("noise", 0.0) > 0.0);
  if (n_osc == 0) p.oscillator_mix = 0.0f;
  else if (n_osc == 1 || (mix1 == 0.0 && mix2 == 0.0)) p.oscillator_mix = 1.0f;
  else p.oscillator_mix = (float)(mix1 / (mix1 + mix2));
  p.amp_envelope = parse_envelope(patch.get("amp-envelope"));
  p.amp_envelope.release = p.amp_envelope.decay; // quirk, patches.rs:133-138
  p.filter_envelope = parse_envelope(patch.get("filter-envelope"));
  p.filter_envelope.release = p.filter_envelope.decay; // patches.rs:154-159
  // LFO (patches.rs:139-145, 269-314)
  const json5::Value* lfo = patch.get("lfo");
  float lfo_duty;
  parse_waveform(lfo ? lfo->get("waveform") : nullptr, p.lfo_waveform, lfo_duty);
  p.lfo_frequency = lfo ? lfo->number_or("frequency", 0.0) : 0.0;
  std::string routing = lfo ? lfo->string_or("routing", "none") : "none";
  if (routing == "none") p.lfo_routing = GROOVE_LFO_NONE;
  else if (routing == "amplitude") p.lfo_routing = GROOVE_LFO_AMPLITUDE;
  else if (routing == "pitch") p.lfo_routing = GROOVE_LFO_PITCH;
  else if (routing == "pulse-width") p.lfo_routing = GROOVE_LFO_PULSE_WIDTH;
  else if (routing == "filter-cutoff") p.lfo_routing = GROOVE_LFO_FILTER_CUTOFF;
  // the shipped patches also carry routings LfoRoutingType does not know yet (docs/DSP_SPEC.md §6)
  else if (routing == "pitch-osc2") p.lfo_routing = GROOVE_LFO_PITCH_OSC2;
  else if (routing == "pw-osc1") p.lfo_routing = GROOVE_LFO_PW_OSC1;
  else if (routing == "pw-osc2") p.lfo_routing = GROOVE_LFO_PW_OSC2;
  else if (routing == "resonance") p.lfo_routing = GROOVE_LFO_RESONANCE;
  else if (routing == "cutoff-amp") p.lfo_routing = GROOVE_LFO_CUTOFF_AMP;
  else { // not a routing at all (one shipped file carries a spreadsheet note in this field)
    p.lfo_routing = GROOVE_LFO_NONE;
    if (warnings) warnings->push_back("lfo routing '" + routing + "' is not a routing; the LFO is left unrouted");
  }
  p.lfo_depth = 0.0f;
  if (lfo) {
    const json5::Value* d = lfo->get("depth");
    if (d && d->is_object() && !d->obj.empty()) {
      const std::string& k = d->obj[0].first;
      const double x = d->obj[0].second->num;
      if (k == "pct") p.lfo_depth = (float)clamp01(x);
      else if (k == "cents") p.lfo_depth = (float)clamp01(1.0 - semis_and_cents(0, x)); // Normal::new clamps (patches.rs:309-311)
    }
  }
  const json5::Value* f24 = patch.get("filter-type-24db");
  const json5::Value* f12 = patch.get("filter-type-12db");
  p.filter_cutoff_hz = (float)(f24 ? f24->number_or("cutoff-hz", 0.0) : 0.0);
  p.filter_passband_ripple = (float)denormalize_q(patch.number_or("filter-resonance", 0.0));
  const double c12 = f12 ? f12->number_or("cutoff-hz", 0.0) : 0.0;
  p.filter_cutoff_start = (float)clamp01(c12 > 0.0 ? frequency_to_percent(c12) : 0.0);
  p.filter_cutoff_end = (float)patch.number_or("filter-envelope-weight", 0.0);
  p.dca_gain = 1.0f;
  p.dca_pan = 0.0f;
  return p;
}

groove_welsh_params welsh_params_from_raw_json(const json5::Value& params, std::vector<std::string>* warnings) {
  const json5::Value* vp = params.get("voice");
  const json5::Value& v = vp && vp->is_object() ? *vp : params;
  groove_welsh_params p{};
  auto osc = [&](const char* name, groove_oscillator_params& o) {
    const json5::Value* x = v.get(name);
    parse_waveform(x ? x->get("waveform") : nullptr, o.waveform, o.duty);
    o.tune = 1.0; o.fixed_hz = 0.0;
    if (x) {
      const json5::Value* t = x->get("frequency-tune");
      if (!t) t = x->get("tune");
      const int note = parse_tune(t, o.tune);
      if (note >= 0) o.fixed_hz = note_to_frequency(note);
      const double fixed = x->number_or("fixed-frequency", 0.0);
      if (fixed > 0.0) o.fixed_hz = fixed;
    }
  };
  osc("oscillator-1", p.oscillator_1);
  osc("oscillator-2", p.oscillator_2);
  p.oscillator_2_sync = v.bool_or("oscillator-2-sync", false) ? 1 : 0;
  p.oscillator_mix = (float)clamp01(v.number_or("oscillator-mix", 1.0));
  p.amp_envelope = parse_envelope(v.get("amp-envelope"));
  p.filter_envelope = parse_envelope(v.get("filter-envelope"));
  const json5::Value* lfo = v.get("lfo");
  float lfo_duty;
  parse_waveform(lfo ? lfo->get("waveform") : nullptr, p.lfo_waveform, lfo_duty);
  p.lfo_frequency = lfo ? lfo->number_or("frequency", 0.0) : 0.0;
  const std::string routing = v.string_or("lfo-routing", "none");
  static const std::pair<const char*, uint32_t> routes[] = {
      {"none", GROOVE_LFO_NONE}, {"amplitude", GROOVE_LFO_AMPLITUDE}, {"pitch", GROOVE_LFO_PITCH}, {"pulse-width", GROOVE_LFO_PULSE_WIDTH},
      {"filter-cutoff", GROOVE_LFO_FILTER_CUTOFF}};   // LfoRouting (patches.rs:271-290)
  p.lfo_routing = GROOVE_LFO_NONE;
  bool known = false;
  for (auto& r : routes) if (routing == r.first) { p.lfo_routing = r.second; known = true; }
  if (!known && warnings) warnings->push_back("lfo-routing '" + routing + "' is not a routing; the LFO is left unrouted");
  p.lfo_depth = (float)clamp01(v.number_or("lfo-depth", 0.0));
  const json5::Value* f = v.get("filter");
  p.filter_cutoff_hz = (float)(f ? f->number_or("cutoff", 0.0) : 0.0);
  p.filter_passband_ripple = (float)(f ? f->number_or("passband-ripple", 0.707) : 0.707);
  p.filter_cutoff_start = (float)clamp01(v.number_or("filter-cutoff-start", 0.0));
  p.filter_cutoff_end = (float)clamp01(v.number_or("filter-cutoff-end", 0.0));
  const json5::Value* dca = params.get("dca");
  if (!dca) dca = v.get("dca");
  p.dca_gain = (float)(dca ? dca->number_or("gain", 1.0) : 1.0);
  p.dca_pan = (float)(dca ? dca->number_or("pan", 0.0) : 0.0);
  return p;
}

namespace {

void parse_effect(const std::string& kind, const json5::Value& v, ProjectDesc::Device& d, std::vector<std::string>& warnings) {
  groove_fx_params& f = d.fx;
  f.ceiling = 1.0f; f.bits = 8; f.cutoff_hz = 1000.0f; f.q = 0.707f; f.passband_ripple = 0.707f; f.voices = 4;
  f.delay_seconds = 0.25f; f.attenuation = 0.5f; f.reverb_seconds = 1.0f; f.wet = 1.0f; f.limit_min = 0.0f; f.limit_max = 1.0f;
  f.bandwidth_hz = 500.0f; f.db_gain = 0.0f;
  d.is_effect = true;
  if (kind == "gain") { d.fx_kind = GROOVE_FX_GAIN; f.ceiling = (float)v.number_or("ceiling", 1.0); }
  else if (kind == "mixer") d.fx_kind = GROOVE_FX_MIXER;
  else if (kind == "limiter") { d.fx_kind = GROOVE_FX_LIMITER; f.limit_min = (float)v.number_or("min", 0.0); f.limit_max = (float)v.number_or("max", 1.0); }
  else if (kind == "compressor") { d.fx_kind = GROOVE_FX_COMPRESSOR; f.limit_min = (float)v.number_or("threshold", 1.0); f.limit_max = (float)v.number_or("ratio", 1.0); }
  else if (kind == "bitcrusher") { d.fx_kind = GROOVE_FX_BITCRUSHER; f.bits = (uint32_t)json5::to_int(v.number_or("bits", v.number_or("bits-to-crush", 8)), 0, 16, 8); }
  else if (kind == "chorus") {
    d.fx_kind = GROOVE_FX_CHORUS; f.voices = (uint32_t)json5::to_int(v.number_or("voices", 4), 1, 64, 4);
    f.delay_seconds = (float)v.number_or("delay-seconds", v.number_or("delay-factor", 0.25));
  }
  else if (kind == "delay") { d.fx_kind = GROOVE_FX_DELAY; f.delay_seconds = (float)v.number_or("seconds", v.number_or("delay", 0.1)); }
  else if (kind == "reverb") { d.fx_kind = GROOVE_FX_REVERB; f.attenuation = (float)v.number_or("attenuation", 0.5); f.reverb_seconds = (float)v.number_or("seconds", 1.0); }
  else if (kind == "filter-low-pass-12db") { d.fx_kind = GROOVE_FX_BIQUAD_LP12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.q = (float)v.number_or("q", 0.707); }
  else if (kind == "filter-high-pass-12db") { d.fx_kind = GROOVE_FX_BIQUAD_HP12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.q = (float)v.number_or("q", 0.707); }
  else if (kind == "filter-band-pass-12db") { d.fx_kind = GROOVE_FX_BIQUAD_BP12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.bandwidth_hz = (float)v.number_or("bandwidth", 500); }
  else if (kind == "filter-band-stop-12db") { d.fx_kind = GROOVE_FX_BIQUAD_BS12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.bandwidth_hz = (float)v.number_or("bandwidth", 500); }
  else if (kind == "filter-all-pass-12db") { d.fx_kind = GROOVE_FX_BIQUAD_AP12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.q = (float)v.number_or("q", 0.707); }
  else if (kind == "filter-peaking-eq-12db") { d.fx_kind = GROOVE_FX_BIQUAD_PEAK12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.db_gain = (float)v.number_or("db-gain", 0); }
  else if (kind == "filter-low-shelf-12db") { d.fx_kind = GROOVE_FX_BIQUAD_LSHELF12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.db_gain = (float)v.number_or("db-gain", 0); }
  else if (kind == "filter-high-shelf-12db") { d.fx_kind = GROOVE_FX_BIQUAD_HSHELF12; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.db_gain = (float)v.number_or("db-gain", 0); }
  else if (kind == "filter-low-pass-24db") { d.fx_kind = GROOVE_FX_BIQUAD_LP24; f.cutoff_hz = (float)v.number_or("cutoff", 1000); f.passband_ripple = (float)v.number_or("passband-ripple", 0.707); }
  else {
    d.fx_kind = GROOVE_FX_MIXER;
    warnings.push_back("effect '" + kind + "' is not built on the GPU path yet; passing audio through");
  }
}

} // namespace

ProjectDesc parse_project(const std::string& text, const std::string& assets_root) {
  json5::ValuePtr root = json5::parse(text);
  if (!root->is_object()) throw std::runtime_error("project: top level is not an object");
  ProjectDesc p;
  p.title = root->string_or("title", "");
  if (const json5::Value* clock = root->get("clock")) {
    p.bpm = clock->number_or("bpm", 128.0);
    if (const json5::Value* ts = clock->get("time-signature")) {
      if (ts->is_array() && ts->arr.size() == 2) { p.ts_top = (int)json5::to_int(ts->arr[0]->num, 1, 64, 4); p.ts_bottom = (int)json5::to_int(ts->arr[1]->num, 1, 64, 4); }
      else if (ts->is_object()) { p.ts_top = (int)json5::to_int(ts->number_or("top", 4), 1, 64, 4); p.ts_bottom = (int)json5::to_int(ts->number_or("bottom", 4), 1, 64, 4); }
    }
  }
  // devices: [{"instrument": [id, {kind: [midi, params]}]}, {"effect": [id, {kind: params}]}, {"controller": …}]
  if (const json5::Value* devs = root->get("devices")) {
    for (auto& dv : devs->arr) {
      if (!dv->is_object() || dv->obj.empty()) continue;
      const std::string& cls = dv->obj[0].first;
      const json5::Value& pair = *dv->obj[0].second;
      if (!pair.is_array() || pair.arr.size() != 2 || !pair.arr[1]->is_object() || pair.arr[1]->obj.empty()) {
        p.warnings.push_back("malformed device entry skipped");
        continue;
      }
      ProjectDesc::Device d;
      d.id = pair.arr[0]->str;
      d.kind = pair.arr[1]->obj[0].first;
      const json5::Value& body = *pair.arr[1]->obj[0].second;
      if (cls == "instrument") {
        // [midi, params] (instruments.rs:26-39: tuple variants); the FM demos of the generation before (projects/demos/instruments/
        // fm-synthesizer-beta-*.json) carry ONE object, {"midi-in": .., "voice": {..}}: taken as the same two things
        // (and projects/tests/load-stereo-wav.json one object with the parameters beside "midi-in")
        const bool flat = body.is_array() && body.arr.size() == 1 && body.arr[0]->is_object();
        if (!flat && (!body.is_array() || body.arr.size() != 2)) { p.warnings.push_back("malformed instrument " + d.id); continue; }
        d.midi_in = (int)json5::to_int(body.arr[0]->number_or("midi-in", 0), 0, 255, 0);
        const json5::Value* voice = flat ? body.arr[0]->get("voice") : nullptr;
        const json5::Value& params = flat ? (voice && voice->is_object() ? *voice : *body.arr[0]) : *body.arr[1];
        if (d.kind == "welsh") {
          d.name = params.string_or("name", "");
          const std::string path = assets_root + "/patches/welsh/" + patch_name_to_settings_name(d.name) + ".json";
          json5::ValuePtr patch = json5::parse(slurp(path)); // the reference panics when the file is missing (patches.rs:84)
          d.welsh = welsh_params_from_patch_json(*patch, &p.warnings);
        } else if (d.kind == "drumkit") {
          d.name = params.string_or("name", "707");
        } else if (d.kind == "sampler") {
          d.name = params.string_or("filename", "");
          d.root = params.number_or("root", 0.0);
        } else if (d.kind == "welsh-raw") {
          d.welsh = welsh_params_from_raw_json(params, &p.warnings);
        } else if (d.kind == "toy-instrument") {
          d.toy_value = params.number_or("fake-value", 0.0);
          if (const json5::Value* dca = params.get("dca")) { d.dca_gain = dca->number_or("gain", 1.0); d.dca_pan = dca->number_or("pan", 0.0); }
        } else if (d.kind == "fm-synthesizer") {
          d.fm.ratio = params.number_or("ratio", 2.0);
          d.fm.depth = (float)params.number_or("depth", 1.0);
          d.fm.beta = (float)params.number_or("beta", 1.0);
          d.fm.carrier_envelope = parse_envelope(params.get("carrier-envelope"));
          d.fm.modulator_envelope = parse_envelope(params.get("modulator-envelope"));
          d.fm.dca_gain = 1.0f; d.fm.dca_pan = 0.0f;
        } else {
          p.warnings.push_back("instrument kind '" + d.kind + "' is not built on the GPU path; skipped");
          continue;
        }
      } else if (cls == "effect") {
        parse_effect(d.kind, body, d, p.warnings);
      } else if (cls == "controller") {
        p.warnings.push_back("controller device '" + d.id + "' (" + d.kind + ") skipped: only tracks and trips drive this path");
        continue;
      } else { // DeviceSettings has three variants (settings/src/lib.rs:42-46); projects/tests/invalid-project.json "should fail to load" on a fourth
        throw std::runtime_error("project: unknown device class '" + cls + "' (expected instrument, controller or effect)");
      }
      p.devices.push_back(d);
    }
  }
  if (const json5::Value* pcs = root->get("patch-cables"))
    for (auto& pc : pcs->arr) {
      std::vector<std::string> ids;
      for (auto& id : pc->arr) ids.push_back(id->str);
      if (ids.size() < 2) { p.warnings.push_back("ignoring patch cable with only one ID"); continue; } // songs.rs:136-139
      p.patch_cables.push_back(ids);
    }
  // patterns + tracks → absolute note events (PatternProgrammer::insert_pattern_at_cursor, songs.rs:210-249)
  struct Pattern { double note_beats; std::vector<std::vector<int>> rows; };
  std::map<std::string, Pattern> patterns;
  if (const json5::Value* pats = root->get("patterns"))
    for (auto& pv : pats->arr) {
      Pattern pat;
      const json5::Value* nv = pv->get("note-value");
      pat.note_beats = nv && nv->is_string() ? beat_value_beats(nv->str) : 4.0 / p.ts_bottom;
      if (const json5::Value* notes = pv->get("notes"))
        for (auto& row : notes->arr) {
          std::vector<int> r;
          for (auto& n : row->arr) r.push_back((int)json5::to_int(n->num, 0, 127, 0)); // 0 = rest
          pat.rows.push_back(r);
        }
      const std::string id = pv->string_or("id", "");
      if (patterns.count(id)) { p.warnings.push_back("duplicate pattern ID " + id + "; skipping all but one"); continue; }
      patterns[id] = pat;
    }
  const double beats_per_measure = p.ts_top * 4.0 / p.ts_bottom;
  if (const json5::Value* tracks = root->get("tracks"))
    for (auto& tv : tracks->arr) {
      const int channel = (int)json5::to_int(tv->number_or("midi-channel", 0), 0, 255, 0);
      double cursor = 0.0;
      if (const json5::Value* ids = tv->get("patterns"))
        for (auto& idv : ids->arr) {
          auto it = patterns.find(idv->str);
          if (it == patterns.end()) continue;
          const Pattern& pat = it->second;
          size_t longest = 0;
          for (auto& row : pat.rows) {
            longest = std::max(longest, row.size());
            for (size_t i = 0; i < row.size(); ++i)
              if (row[i] != 0) p.notes.push_back({channel, row[i], cursor + i * pat.note_beats, pat.note_beats});
          }
          const double len = longest * pat.note_beats;
          cursor += std::ceil(len / beats_per_measure - 1e-9) * beats_per_measure; // whole measures
        }
      p.end_beats = std::max(p.end_beats, cursor);
    }
  // paths + trips (songs.rs:251-306; ControlTripSettings has no start field at this commit: trips start at 0)
  struct Path { double step_beats; std::vector<ControlStep> steps; };
  std::map<std::string, Path> paths;
  if (const json5::Value* pv = root->get("paths"))
    for (auto& x : pv->arr) {
      Path path;
      const json5::Value* nv = x->get("note-value");
      path.step_beats = nv && nv->is_string() ? beat_value_beats(nv->str) : 4.0 / p.ts_bottom;
      if (const json5::Value* steps = x->get("steps"))
        for (auto& sv : steps->arr) {
          if (!sv->is_object() || sv->obj.empty()) continue;
          const std::string& k = sv->obj[0].first;
          const json5::Value& b = *sv->obj[0].second;
          ControlStep st;
          st.beats = path.step_beats;
          if (k == "flat") { st.kind = ControlStep::FLAT; st.start = st.end = b.number_or("value", 0.0); }
          else if (k == "slope") { st.kind = ControlStep::SLOPE; st.start = b.number_or("start", 0); st.end = b.number_or("end", 0); }
          else if (k == "logarithmic") { st.kind = ControlStep::LOGARITHMIC; st.start = b.number_or("start", 0); st.end = b.number_or("end", 0); }
          else if (k == "exponential") { st.kind = ControlStep::EXPONENTIAL; st.start = b.number_or("start", 0); st.end = b.number_or("end", 0); }
          else { st.kind = ControlStep::TRIGGERED; }
          path.steps.push_back(st);
        }
      paths[x->string_or("id", "")] = path;
    }
  if (const json5::Value* tv = root->get("trips"))
    for (auto& x : tv->arr) {
      ProjectDesc::Trip t;
      t.id = x->string_or("id", "");
      if (const json5::Value* tg = x->get("target")) { t.target = tg->string_or("id", ""); t.param = tg->string_or("param", ""); }
      if (const json5::Value* ids = x->get("paths"))
        for (auto& idv : ids->arr) {
          auto it = paths.find(idv->str);
          if (it == paths.end()) { p.warnings.push_back("trip " + t.id + " refers to nonexistent path " + idv->str); continue; }
          t.steps.insert(t.steps.end(), it->second.steps.begin(), it->second.steps.end());
        }
      p.trips.push_back(t);
    }
  return p;
}

ProjectDesc parse_project_file(const std::string& path, const std::string& assets_root) {
  ProjectDesc p = parse_project(slurp(path), assets_root);
  const size_t slash = path.find_last_of('/');
  p.project_dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
  return p;
}

std::string describe(const ProjectDesc& p) {
  std::ostringstream o;
  o.precision(10);
  o << "{\"title\":\"" << p.title << "\",\"bpm\":" << p.bpm << ",\"time_signature\":[" << p.ts_top << "," << p.ts_bottom << "],\"devices\":[";
  for (size_t i = 0; i < p.devices.size(); ++i) {
    const auto& d = p.devices[i];
    o << (i ? "," : "") << "{\"id\":\"" << d.id << "\",\"kind\":\"" << d.kind << "\",\"effect\":" << (d.is_effect ? "true" : "false")
      << ",\"midi_in\":" << d.midi_in << ",\"name\":\"" << d.name << "\",\"fx_kind\":" << d.fx_kind << ",\"cutoff\":" << d.fx.cutoff_hz
      << ",\"passband_ripple\":" << d.fx.passband_ripple << ",\"welsh_osc1\":" << d.welsh.oscillator_1.waveform
      << ",\"welsh_mix\":" << d.welsh.oscillator_mix << ",\"welsh_cutoff\":" << d.welsh.filter_cutoff_hz
      << ",\"welsh_release\":" << d.welsh.amp_envelope.release << ",\"welsh_routing\":" << d.welsh.lfo_routing << "}";
  }
  o << "],\"patch_cables\":[";
  for (size_t i = 0; i < p.patch_cables.size(); ++i) {
    o << (i ? "," : "") << "[";
    for (size_t j = 0; j < p.patch_cables[i].size(); ++j) o << (j ? "," : "") << "\"" << p.patch_cables[i][j] << "\"";
    o << "]";
  }
  o << "],\"n_notes\":" << p.notes.size() << ",\"end_beats\":" << p.end_beats << ",\"trips\":[";
  for (size_t i = 0; i < p.trips.size(); ++i) {
    double beats = 0; for (auto& s : p.trips[i].steps) beats += s.beats;
    o << (i ? "," : "") << "{\"id\":\"" << p.trips[i].id << "\",\"target\":\"" << p.trips[i].target << "\",\"param\":\"" << p.trips[i].param
      << "\",\"steps\":" << p.trips[i].steps.size() << ",\"beats\":" << beats << ",\"first_kind\":" << (p.trips[i].steps.empty() ? -1 : (int)p.trips[i].steps[0].kind) << "}";
  }
  o << "],\"warnings\":" << p.warnings.size() << "}";
  return o.str();
}

bool read_wav_mono(const std::string& path, std::vector<float>& out, uint32_t* sample_rate, std::string* err) {
  std::string data;
  try { data = slurp(path); } catch (const std::exception& e) { if (err) *err = e.what(); return false; }
  auto rd16 = [&](size_t o) { return (uint32_t)(uint8_t)data[o] | ((uint32_t)(uint8_t)data[o + 1] << 8); };
  auto rd32 = [&](size_t o) { return rd16(o) | (rd16(o + 2) << 16); };
  if (data.size() < 12 || data.compare(0, 4, "RIFF") != 0 || data.compare(8, 4, "WAVE") != 0) { if (err) *err = "not a RIFF/WAVE file: " + path; return false; }
  uint32_t fmt = 0, channels = 0, rate = 0, bits = 0;
  size_t pos = 12, dpos = 0, dlen = 0;
  while (pos + 8 <= data.size()) {
    const std::string id = data.substr(pos, 4);
    const size_t len = rd32(pos + 4);
    if (id == "fmt " && pos + 8 + 16 <= data.size()) { fmt = rd16(pos + 8); channels = rd16(pos + 10); rate = rd32(pos + 12); bits = rd16(pos + 22);
      if (fmt == 0xFFFE && len >= 26 && pos + 8 + 26 <= data.size()) fmt = rd16(pos + 8 + 24); }
    else if (id == "data") { dpos = pos + 8; dlen = std::min(len, data.size() - dpos); }
    pos += 8 + len + (len & 1);
  }
  if (!dpos || !channels || !bits) { if (err) *err = "WAV without fmt/data chunk: " + path; return false; }
  if ((bits != 8 && bits != 16 && bits != 24 && bits != 32) || channels > 64) {
    if (err) *err = "unsupported WAV format (" + std::to_string(bits) + " bits, " + std::to_string(channels) + " channels): " + path;
    return false;
  }
  const size_t bytes = bits / 8, frames = dlen / (bytes * channels);
  out.resize(frames);
  for (size_t f = 0; f < frames; ++f) {
    double acc = 0.0;
    for (uint32_t c = 0; c < channels; ++c) {
      const size_t o = dpos + (f * channels + c) * bytes;
      double v = 0.0;
      if (fmt == 3 && bits == 32) { float x; std::memcpy(&x, &data[o], 4); v = x; }
      else if (bits == 16) v = (int16_t)rd16(o) / 32768.0;                          // ints scaled by 2^(bits-1), A.10
      else if (bits == 24) { int32_t x = (int32_t)(rd16(o) | ((uint32_t)(uint8_t)data[o + 2] << 16)); if (x & 0x800000) x |= ~0xFFFFFF; v = x / 8388608.0; }
      else if (bits == 32) v = (int32_t)rd32(o) / 2147483648.0;
      else if (bits == 8) v = ((uint8_t)data[o] - 128) / 128.0;
      acc += v;
    }
    out[f] = (float)(acc / channels);
  }
  if (sample_rate) *sample_rate = rate;
  return true;
}

int read_wav_root_note(const std::string& path) {
  std::string data;
  try { data = slurp(path); } catch (const std::exception&) { return -1; }
  auto rd16 = [&](size_t o) { return (uint32_t)(uint8_t)data[o] | ((uint32_t)(uint8_t)data[o + 1] << 8); };
  auto rd32 = [&](size_t o) { return rd16(o) | (rd16(o + 2) << 16); };
  if (data.size() < 12 || data.compare(0, 4, "RIFF") != 0 || data.compare(8, 4, "WAVE") != 0) return -1;
  int smpl = -1, acid = -1;
  size_t pos = 12;
  while (pos + 8 <= data.size()) {
    const std::string id = data.substr(pos, 4);
    const size_t len = rd32(pos + 4);
    // smpl: manufacturer, product, sample period, MIDI unity note, pitch fraction, ... (unity note at byte 12 of the chunk)
    if (id == "smpl" && len >= 16 && pos + 8 + 16 <= data.size()) { const uint32_t n = rd32(pos + 8 + 12); if (n <= 127) smpl = (int)n; }
    // acid: flags (bit 1: the root note is set), root note (u16), ...
    else if (id == "acid" && len >= 6 && pos + 8 + 6 <= data.size()) { const uint32_t fl = rd32(pos + 8), n = rd16(pos + 8 + 4); if ((fl & 2u) && n <= 127) acid = (int)n; }
    if (len > data.size()) break;
    pos += 8 + len + (len & 1);
  }
  return smpl >= 0 ? smpl : acid;
}

namespace {
// Drumkit "707": GM percussion key → sample file (Appendix A.10).
const std::pair<int, const char*> k707[] = {
    {35, "Kick 1 R1.wav"}, {36, "Kick 2 R1.wav"}, {37, "Rim R1.wav"}, {38, "Snare 1 R1.wav"}, {39, "Clap R1.wav"},
    {40, "Snare 2 R1.wav"}, {41, "Tom 1 R1.wav"}, {42, "Hat Closed R1.wav"}, {43, "Tom 1 R1.wav"}, {44, "Hat Closed R1.wav"},
    {45, "Tom 2 R1.wav"}, {46, "Hat Open R1.wav"}, {47, "Tom 2 R1.wav"}, {48, "Tom 3 R1.wav"}, {49, "Crash R1.wav"},
    {50, "Tom 3 R1.wav"}, {51, "Ride R1.wav"}, {54, "Tambourine R1.wav"}, {56, "Cowbell R1.wav"}};

// --synthetic-kit: a decaying tone + noise per drum key, deterministic; stands in for the CC0 707 samples
// on boxes without the assets directory (the GPU box, the tests).
void synthetic_707_kit(uint32_t sample_rate, std::vector<float>& pcm, std::vector<groove_sample_desc>& descs, int key_to_sample[128]) {
  pcm.clear(); descs.clear();
  for (int k = 0; k < 128; ++k) key_to_sample[k] = -1;
  for (auto& km : k707) {
    const uint32_t len = 12000 + 900 * (uint32_t)(km.first % 13);
    std::vector<float> one(len);
    uint32_t lcg = 12345u + (uint32_t)km.first;
    for (uint32_t i = 0; i < len; ++i) {
      lcg = lcg * 1664525u + 1013904223u;
      const double env = std::exp(-(double)i / (0.08 * sample_rate));
      one[i] = (float)(env * (0.7 * std::sin(2.0 * 3.14159265358979 * (50.0 + 4.0 * km.first) * i / sample_rate) +
                              0.2 * ((double)(lcg >> 8) / 8388608.0 - 1.0)));
    }
    groove_sample_desc sd{(uint64_t)pcm.size(), (uint32_t)one.size(), 0.0f};
    key_to_sample[km.first] = (int)descs.size();
    descs.push_back(sd);
    pcm.insert(pcm.end(), one.begin(), one.end());
  }
}
#ifndef GROOVE_HOST_PARSER_ONLY
extern "C" int gh_add_drumkit(void*, const float*, uint64_t, const groove_sample_desc*, uint32_t, const int*);
extern "C" int gh_add_welsh(void*, const groove_welsh_params*, uint32_t);
extern "C" int gh_add_fm(void*, const groove_fm_params*, uint32_t);
extern "C" int gh_add_sampler(void*, const float*, uint64_t, double, uint32_t);
extern "C" int gh_add_toy_instrument(void*, double, double);
// Where a sampler's file is looked for (the reference resolves it through groove_utils::Paths, absent from the tree): the assets'
// samples directory, the assets directory, the project's own directory, the reference checkout's test-data (where the three
// sampler projects' files lie: projects/tests/load-*-wav.json, projects/demos/instruments/sampler.json).
std::string find_sample_file(const std::string& name, const std::string& assets_root, const std::string& project_dir) {
  if (name.empty()) return name;
  std::vector<std::string> tries;
  if (name[0] == '/') tries.push_back(name);
  if (!assets_root.empty()) { tries.push_back(assets_root + "/samples/" + name); tries.push_back(assets_root + "/" + name); }
  if (!project_dir.empty()) tries.push_back(project_dir + "/" + name);
  if (!assets_root.empty()) { tries.push_back(assets_root + "/../test-data/" + name); tries.push_back(assets_root + "/../test-data/samples/" + name); }
  for (const auto& t : tries) { std::ifstream f(t, std::ios::binary); if (f) return t; }
  return tries.empty() ? name : tries.front();
}
#endif
} // namespace

#ifndef GROOVE_HOST_PARSER_ONLY // (the sanitizer build of the parser half — parse_check.cpp, `make asan` — needs no device library)
int instantiate(Orchestrator& o, const ProjectDesc& p, const std::string& assets_root, bool synthetic_kit) {
  std::map<std::string, Uid> uid_of;
  uid_of["main-mixer"] = kMainMixerUid;
  o.set_bpm(p.bpm);
  for (const auto& d : p.devices) {
    int uid = -1;
    if (d.is_effect) {
      uid = (int)o.add(std::unique_ptr<Entity>(new FxEffect(o.ctx(), d.fx_kind, &d.fx, 1)));
    } else if (d.kind == "welsh") {
      uid = gh_add_welsh(&o, &d.welsh, 8); // voice store: 8 voices per synth (Appendix A.6)
    } else if (d.kind == "fm-synthesizer") {
      uid = gh_add_fm(&o, &d.fm, 8);
    } else if (d.kind == "drumkit") {
      std::vector<float> pcm;
      std::vector<groove_sample_desc> descs;
      int key_to_sample[128];
      for (int& k : key_to_sample) k = -1;
      if (synthetic_kit) {
        synthetic_707_kit(o.sample_rate(), pcm, descs, key_to_sample);
      } else {
        for (auto& km : k707) {
          std::vector<float> one;
          std::string err;
          uint32_t sr = 0;
          if (!read_wav_mono(assets_root + "/samples/elphnt.io/707/" + km.second, one, &sr, &err)) return o.fail(err);
          groove_sample_desc sd{(uint64_t)pcm.size(), (uint32_t)one.size(), 0.0f};
          key_to_sample[km.first] = (int)descs.size();
          descs.push_back(sd);
          pcm.insert(pcm.end(), one.begin(), one.end());
        }
      }
      uid = gh_add_drumkit(&o, pcm.data(), pcm.size(), descs.data(), (uint32_t)descs.size(), key_to_sample);
    } else if (d.kind == "welsh-raw") {
      uid = gh_add_welsh(&o, &d.welsh, 8);
    } else if (d.kind == "toy-instrument") {
      uid = gh_add_toy_instrument(&o, d.dca_gain, d.dca_pan);
    } else if (d.kind == "sampler") {
      // SamplerParams{filename, root} (instruments.rs:81-88): root > 0 is the sample's pitch; 0 = the file's own root note (smpl /
      // acid chunk), else 440 Hz — projects/tests/load-mono-wav.json plays a spoken sentence with root 0 on key 69 (docs/DSP_SPEC.md section 7)
      const std::string path = find_sample_file(d.name, assets_root, p.project_dir);
      std::vector<float> pcm;
      std::string err;
      uint32_t sr = 0;
      if (!read_wav_mono(path, pcm, &sr, &err)) return o.fail(err);
      if (pcm.empty()) return o.fail("sampler: no frames in " + path);
      double root = d.root;
      if (!(root > 0.0)) { const int note = read_wav_root_note(path); root = note >= 0 ? note_to_frequency(note) : 440.0; }
      uid = gh_add_sampler(&o, pcm.data(), pcm.size(), root, 8);
    } else {
      return o.fail("instrument kind '" + d.kind + "' is not an instrument kind (settings/src/instruments.rs:26-39)");
    }
    if (uid < 0) return 1;
    o.get((Uid)uid)->name = d.id;
    uid_of[d.id] = (Uid)uid;
    if (!d.is_effect && o.connect_midi_downstream((Uid)uid, (uint8_t)d.midi_in)) return 1;
  }
  for (const auto& cable : p.patch_cables)
    for (size_t i = 0; i + 1 < cable.size(); ++i) {
      auto a = uid_of.find(cable[i]), b = uid_of.find(cable[i + 1]);
      if (a == uid_of.end() || b == uid_of.end()) continue; // "Warning: … patch ID not found" (songs.rs:152-156)
      if (o.patch(a->second, b->second)) return 1;
    }
  if (!p.notes.empty() || p.end_beats > 0.0) {
    auto seq = std::unique_ptr<Sequencer>(new Sequencer());
    for (const auto& n : p.notes) seq->insert((uint8_t)n.channel, (uint8_t)n.key, n.start_beat, n.duration_beats);
    seq->set_end_beats(p.end_beats);
    o.add(std::move(seq));
  }
  for (const auto& t : p.trips) {
    auto target = uid_of.find(t.target);
    if (target == uid_of.end()) continue; // "Warning: trip … controls nonexistent entity" (songs.rs:300-304)
    Entity* e = o.get(target->second);
    if (!e || !(e->is_effect() || e->is_instrument())) continue;
    // Controllable is generated for instruments too (proc-macros/src/control.rs:171-183): the trip's values reach the synth's voices
    // through groove_bank_set_param (round 6; trips onto instruments were dropped before)
    const int idx = e->is_effect() ? static_cast<Effect*>(e)->control_index_for_name(t.param) : static_cast<Instrument*>(e)->control_index_for_name(t.param);
    if (idx < 0) continue; // "trip … not added because of error" (songs.rs:292-297)
    auto trip = std::unique_ptr<ControlTrip>(new ControlTrip(target->second, (uint32_t)idx, t.start_beat));
    for (const auto& s : t.steps) trip->add_step(s);
    o.add(std::move(trip));
  }
  return 0;
}

#endif // GROOVE_HOST_PARSER_ONLY
} // namespace groove_host

// ---- C surface ------------------------------------------------------------------------------
using namespace groove_host;
extern "C" {
// read_wav_mono for the tests: frames written (<= cap) or -1 with a message.
int64_t gh_read_wav_mono(const char* path, float* out, uint64_t cap, uint32_t* sample_rate, char* err, size_t err_cap) {
  std::vector<float> v;
  std::string e;
  if (!groove_host::read_wav_mono(path, v, sample_rate, &e)) {
    if (err && err_cap) { std::snprintf(err, err_cap, "%s", e.c_str()); }
    return -1;
  }
  const uint64_t n = std::min<uint64_t>(cap, v.size());
  if (n && out) std::memcpy(out, v.data(), n * sizeof(float));
  return (int64_t)v.size();
}

// Parse only (no GPU): returns a malloc'ed JSON summary, or NULL and the message in err[0..err_len).
char* gh_project_describe(const char* path, const char* assets_root, char* err, size_t err_len) {
  try {
    const std::string s = describe(parse_project_file(path, assets_root ? assets_root : ""));
    char* out = (char*)std::malloc(s.size() + 1);
    std::memcpy(out, s.c_str(), s.size() + 1);
    return out;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return nullptr;
  }
}
char* gh_project_describe_text(const char* text, const char* assets_root, char* err, size_t err_len) {
  try {
    const std::string s = describe(parse_project(text, assets_root ? assets_root : ""));
    char* out = (char*)std::malloc(s.size() + 1);
    std::memcpy(out, s.c_str(), s.size() + 1);
    return out;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return nullptr;
  }
}
void gh_free(void* p) { std::free(p); }
// The root note a WAV file carries (smpl chunk, else acid chunk), or -1.
int gh_read_wav_root_note(const char* path) { return groove_host::read_wav_root_note(path ? path : ""); }
// Welsh patch JSON text → groove_welsh_params (no GPU).
int gh_welsh_params_from_patch_json(const char* text, groove_welsh_params* out, char* err, size_t err_len) {
  try {
    *out = welsh_params_from_patch_json(*json5::parse(text), nullptr);
    return 0;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return 1;
  }
}
// The --synthetic-kit sample bank as data (no GPU): pcm_out may be NULL to ask for the sizes only.
int gh_synthetic_kit(uint32_t sample_rate, float* pcm_out, uint64_t pcm_cap, groove_sample_desc* descs_out, uint32_t descs_cap,
                     int* key_to_sample /*[128]*/, uint64_t* pcm_frames, uint32_t* n_descs) {
  std::vector<float> pcm;
  std::vector<groove_sample_desc> descs;
  int k2s[128];
  synthetic_707_kit(sample_rate, pcm, descs, k2s);
  if (pcm_frames) *pcm_frames = pcm.size();
  if (n_descs) *n_descs = (uint32_t)descs.size();
  if (!pcm_out) return 0;
  if (pcm_cap < pcm.size() || descs_cap < descs.size() || !descs_out || !key_to_sample) return 1;
  std::memcpy(pcm_out, pcm.data(), pcm.size() * sizeof(float));
  std::memcpy(descs_out, descs.data(), descs.size() * sizeof(groove_sample_desc));
  std::memcpy(key_to_sample, k2s, sizeof(k2s));
  return 0;
}
#ifndef GROOVE_HOST_PARSER_ONLY
// Load a project into an existing orchestrator (GPU).
int gh_load_project(void* h, const char* path, const char* assets_root, int synthetic_kit) {
  Orchestrator* o = (Orchestrator*)h;
  try {
    ProjectDesc p = parse_project_file(path, assets_root ? assets_root : "");
    return instantiate(*o, p, assets_root ? assets_root : "", synthetic_kit != 0);
  } catch (const std::exception& e) {
    return o->fail(e.what());
  }
}
#endif
}
