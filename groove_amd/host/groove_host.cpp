// groove_host.cpp — see groove_host.hpp.  Every audio operation goes through the C ABI of
// libgroove_hip.so; there is no CPU audio path here.
#include "groove_host.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace groove_host {

// ------------------------------------------------------------------ VoiceBankInstrument
VoiceBankInstrument::VoiceBankInstrument(groove_ctx* ctx, groove_bank* bank, uint32_t voices, bool sum_voices,
                                         double release_seconds, bool one_voice_per_key)
    : ctx_(ctx), bank_(bank), voices_(voices), sum_voices_(sum_voices), release_seconds_(release_seconds),
      per_key_(one_voice_per_key), key_of_voice_(voices, -1), busy_until_(voices, 0), started_(voices, 0) {
  groove_block_create(ctx_, voices_, GROOVE_BLOCK_FRAMES, &block_);
  if (sum_voices_) groove_block_create(ctx_, 1, GROOVE_BLOCK_FRAMES, &summed_);
  short_render_ = !per_key_ && voices_ <= groove_time_parallel_max_voices(ctx_); // synth banks of this size render time-parallel
}
VoiceBankInstrument::~VoiceBankInstrument() {
  if (summed_) groove_block_destroy(summed_);
  if (block_next_) groove_block_destroy(block_next_);
  if (block_) groove_block_destroy(block_);
  if (bank_) groove_bank_destroy(bank_);
}
int VoiceBankInstrument::tick(uint32_t frames) {
  if (groove_bank_render(bank_, frames, block_)) return 1;
  if (sum_voices_) return groove_block_accumulate(summed_, block_, frames, 0); // Synthesizer: sum of voices
  return 0;
}
int VoiceBankInstrument::render_ahead(uint32_t frames) {
  if (!block_next_ && groove_block_create(ctx_, voices_, GROOVE_BLOCK_FRAMES, &block_next_)) return 1;
  return groove_bank_render_async(bank_, frames, block_next_); // side streams; consumers of the block wait for it
}
int VoiceBankInstrument::finish(uint32_t frames) {
  if (!block_next_) return tick(frames); // nothing was rendered ahead
  std::swap(block_, block_next_);
  if (sum_voices_) return groove_block_accumulate(summed_, block_, frames, 0);
  return 0;
}
void VoiceBankInstrument::note_on(uint8_t key, uint8_t velocity, uint64_t now) {
  // VoiceStore: first idle voice (Appendix A.6); Drumkit: VoicePerNoteStore (A.10).  "Idle" is
  // tracked on the host: a voice is busy from note-on until note-off + the patch's release time.
  uint32_t v = voices_;
  if (per_key_) {
    for (uint32_t i = 0; i < voices_; ++i) if (key_of_voice_[i] == (int)key) { v = i; break; }
  }
  if (v == voices_)
    for (uint32_t i = 0; i < voices_; ++i)
      if (key_of_voice_[i] < 0 && busy_until_[i] <= now) { v = i; break; }
  if (v == voices_) { // all busy: steal the voice that started first
    v = 0;
    for (uint32_t i = 1; i < voices_; ++i) if (started_[i] < started_[v]) v = i;
  }
  key_of_voice_[v] = key;
  started_[v] = now;
  busy_until_[v] = UINT64_MAX;
  last_voice_ = v;
  groove_note_event e{v, key, velocity, 1, 0};
  groove_bank_note_events(bank_, &e, 1);
}
void VoiceBankInstrument::note_off(uint8_t key, uint8_t velocity, uint64_t now) {
  for (uint32_t i = 0; i < voices_; ++i) {
    if (key_of_voice_[i] == (int)key) {
      groove_note_event e{i, key, velocity, 0, 0};
      groove_bank_note_events(bank_, &e, 1);
      if (!per_key_) key_of_voice_[i] = -1;
      busy_until_[i] = now + (uint64_t)std::ceil(release_seconds_ * groove_sample_rate(ctx_)) + 1;
      if (per_key_) busy_until_[i] = now;
    }
  }
}

int VoiceBankInstrument::control_index_for_name(const std::string& name) const {
  // #[derive(Control)] kebab-case names, nested fields joined with '-' (proc-macros/src/control.rs:127-131, 165): WelshSynth's `dca`
  // field gives dca-gain / dca-pan; the filter's cutoff is the one Welsh voice parameter the reference's demos automate on effects
  static const std::pair<const char*, int> table[] = {
      {"dca-gain", GROOVE_CTL_WELSH_DCA_GAIN}, {"gain", GROOVE_CTL_WELSH_DCA_GAIN}, {"dca-pan", GROOVE_CTL_WELSH_DCA_PAN}, {"pan", GROOVE_CTL_WELSH_DCA_PAN},
      {"filter-cutoff", GROOVE_CTL_WELSH_CUTOFF}, {"cutoff", GROOVE_CTL_WELSH_CUTOFF}};
  for (auto& t : table) if (name == t.first) return t.second;
  return -1;
}
int VoiceBankInstrument::control_set_param(uint32_t index, double value01) {
  return groove_bank_set_param(bank_, GROOVE_ALL_VOICES, index, value01); // (fails, with the library's message, on a bank that has no such control)
}

// ------------------------------------------------------------------ ToyAudioSource
ToyAudioSource::ToyAudioSource(groove_ctx* ctx, double level) : ctx_(ctx), level_((float)level) {
  groove_block_create(ctx_, 1, GROOVE_BLOCK_FRAMES, &block_);
  host_.assign(2 * GROOVE_BLOCK_FRAMES, level_);
  groove_block_upload(block_, host_.data(), GROOVE_BLOCK_FRAMES);
}
ToyAudioSource::~ToyAudioSource() { if (block_) groove_block_destroy(block_); }
int ToyAudioSource::tick(uint32_t) { return 0; } // the block already holds `level` everywhere

// ------------------------------------------------------------------ FxEffect
FxEffect::FxEffect(groove_ctx* ctx, uint32_t kind, const groove_fx_params* p, uint32_t lanes) : ctx_(ctx), lanes_(lanes) {
  groove_fx_create(ctx_, kind, p, lanes, &fx_);
}
FxEffect::~FxEffect() { if (fx_) groove_fx_destroy(fx_); }
int FxEffect::transform_audio(groove_block* inout, uint32_t frames) {
  if (!fx_) return 1;
  return groove_fx_process(fx_, inout, frames);
}
int FxEffect::control_set_param(uint32_t index, double value01) {
  if (!fx_) return 1;
  return groove_fx_set_param(fx_, GROOVE_ALL_VOICES, index, value01);
}
int FxEffect::control_index_for_name(const std::string& name) const {
  // #[derive(Control)] kebab-case names (proc-macros/src/control.rs:127-131, 165)
  static const std::pair<const char*, int> table[] = {
      {"ceiling", GROOVE_CTL_FX_CEILING}, {"bits", GROOVE_CTL_FX_BITS}, {"bits-to-crush", GROOVE_CTL_FX_BITS},
      {"cutoff", GROOVE_CTL_FX_CUTOFF}, {"q", GROOVE_CTL_FX_Q}, {"passband-ripple", GROOVE_CTL_FX_PASSBAND_RIPPLE},
      {"attenuation", GROOVE_CTL_FX_ATTENUATION}, {"wet-dry-mix", GROOVE_CTL_FX_WET}, {"threshold", GROOVE_CTL_FX_THRESHOLD}};
  for (auto& t : table) if (name == t.first) return t.second;
  return -1;
}

// ------------------------------------------------------------------ Sequencer / ControlTrip
void Sequencer::insert(uint8_t channel, uint8_t key, double start_beat, double duration_beats) {
  const uint64_t a = MusicalTime::from_beats(start_beat).units;
  const uint64_t b = MusicalTime::from_beats(start_beat + duration_beats).units;
  // kept ordered by time, equal times in insertion order (what a stable sort after every insert gave,
  // without sorting a large project's whole list once per note)
  auto put = [&](const Ev& e) {
    events_.insert(std::upper_bound(events_.begin(), events_.end(), e, [](const Ev& x, const Ev& y) { return x.at < y.at; }), e);
  };
  put({a, channel, key, true});
  put({b, channel, key, false});
  if (!explicit_end_) end_ = std::max(end_, b);
}
void Sequencer::work(uint64_t start_units, uint64_t end_units, std::vector<MidiEvent>& out, Orchestrator&) {
  for (const Ev& e : events_)
    if (e.at >= start_units && e.at < end_units) out.push_back({e.channel, e.key, 127, e.on});
}
double ControlTrip::value_at(const ControlStep& s, double t) {
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  switch (s.kind) {
    case ControlStep::FLAT: return s.start;
    case ControlStep::SLOPE: return s.start + (s.end - s.start) * t;
    // logarithmic / exponential shapes: the MMA convex / concave transforms of the linear ramp
    // (orchestration/src/util.rs:4-21)
    case ControlStep::LOGARITHMIC: {
      const double c = t < std::pow(10.0, -12.0 / 5.0) ? 0.0 : 1.0 + (5.0 / 12.0) * std::log10(t);
      return s.start + (s.end - s.start) * c;
    }
    case ControlStep::EXPONENTIAL: {
      const double c = t > 1.0 - std::pow(10.0, -12.0 / 5.0) ? 1.0 : -(5.0 / 12.0) * std::log10(1.0 - t);
      return s.start + (s.end - s.start) * c;
    }
    default: return s.start;
  }
}
uint64_t ControlTrip::end_units_() const {
  double b = start_;
  for (auto& s : steps_) b += s.beats;
  return MusicalTime::from_beats(b).units;
}
void ControlTrip::work(uint64_t start_units, uint64_t end_units, std::vector<MidiEvent>&, Orchestrator& o) {
  (void)end_units;
  const double now = (double)start_units / MusicalTime::UNITS_IN_BEAT; // block-granular: value at block start
  double b = start_;
  for (auto& s : steps_) {
    if (now >= b && now < b + s.beats) {
      const double v = value_at(s, (now - b) / s.beats);
      if (v != last_sent_) {
        last_sent_ = v;
        o.control_effect(target_, index_, v);
      }
      return;
    }
    b += s.beats;
  }
}

// ------------------------------------------------------------------ Orchestrator
namespace {
// The main mixer is an effect whose transform_audio is the identity (orchestrator.rs:543-546).
class MainMixer : public Effect {
 public:
  uint32_t lanes() const override { return 1; }
  int transform_audio(groove_block*, uint32_t) override { return 0; }
};
} // namespace

Orchestrator::Orchestrator(int device, uint32_t sample_rate, double bpm) : sr_(sample_rate), bpm_(bpm) {
  if (groove_init(device, &ctx_)) { err_ = groove_last_error(nullptr); ctx_ = nullptr; return; }
  if (sample_rate != GROOVE_DEFAULT_SAMPLE_RATE) groove_update_sample_rate(ctx_, sample_rate);
  nodes_.emplace_back();
  nodes_[0].entity.reset(new MainMixer());
  nodes_[0].entity->uid = 0;
  nodes_[0].entity->name = "main-mixer";
  bus_frames_ = GROOVE_BLOCK_FRAMES;
  groove_bus_create(ctx_, bus_frames_, &bus_);
}
Orchestrator::~Orchestrator() {
  if (!ctx_) return;
  for (auto& n : nodes_) if (n.accum) groove_block_destroy(n.accum);
  nodes_.clear(); // entities free their banks / effects before the context goes away
  if (bus_) groove_bus_destroy(ctx_, bus_);
  groove_shutdown(ctx_);
}
int Orchestrator::update_sample_rate(uint32_t hz) {
  if (groove_update_sample_rate(ctx_, hz)) return fail(groove_last_error(ctx_));
  sr_ = hz;
  return 0;
}
Uid Orchestrator::add(std::unique_ptr<Entity> e) {
  nodes_.emplace_back();
  const Uid uid = nodes_.size() - 1;
  e->uid = uid;
  nodes_.back().entity = std::move(e);
  return uid;
}
Entity* Orchestrator::get(Uid uid) { return uid < nodes_.size() ? nodes_[uid].entity.get() : nullptr; }
int Orchestrator::patch(Uid source, Uid sink) {
  Entity* in = get(sink);
  if (!in) return fail("Couldn't find input_uid");
  if (!in->is_effect()) return fail("Input device doesn't transform audio and can't be patched from output device");
  Entity* out = get(source);
  if (!out) return fail("Couldn't find output_uid");
  if (!(out->is_instrument() || out->is_effect())) return fail("Output device doesn't output audio and can't be patched into input device");
  if (source == sink) return fail("can't patch a device into itself");
  nodes_[sink].sources.push_back(source);
  fanout_valid_ = false;
  return 0;
}
int Orchestrator::patch_chain_to_main_mixer(const std::vector<Uid>& uids) {
  for (size_t i = 0; i + 1 < uids.size(); ++i) if (patch(uids[i], uids[i + 1])) return 1;
  if (!uids.empty()) return patch(uids.back(), kMainMixerUid);
  return 0;
}
void Orchestrator::unpatch_all() { for (auto& n : nodes_) n.sources.clear(); fanout_valid_ = false; }
int Orchestrator::connect_midi_downstream(Uid receiver, uint8_t channel) {
  Entity* e = get(receiver);
  if (!e || !e->is_instrument()) return fail("MIDI receiver is not an instrument");
  midi_receivers_.insert({channel, receiver});
  return 0;
}
int Orchestrator::ensure_accum(Node& n, uint32_t lanes) {
  if (n.accum && n.accum_lanes == lanes) return 0;
  if (n.accum) groove_block_destroy(n.accum);
  n.accum = nullptr;
  if (groove_block_create(ctx_, lanes, GROOVE_BLOCK_FRAMES, &n.accum)) return fail(groove_last_error(ctx_));
  n.accum_lanes = lanes;
  return 0;
}
// Post-order evaluation of one node for one block (the per-frame DFS of gather_audio, run per block).
int Orchestrator::eval(Uid uid, uint32_t frames, groove_block** out_block, uint32_t* out_lanes) {
  Entity* e = nodes_[uid].entity.get();
  if (e->is_instrument()) {
    Instrument* ins = static_cast<Instrument*>(e);
    if (!ahead_eval_ && ins->tick(frames)) return fail(groove_last_error(ctx_));
    *out_block = ins->output();
    *out_lanes = ins->lanes();
    return 0;
  }
  if (!e->is_effect()) { *out_block = nullptr; *out_lanes = 0; return 0; }
  Effect* fx = static_cast<Effect*>(e);
  const uint32_t lanes = fx->lanes();
  // A run of library effects patched one into the next, each heard by nobody else, is what the reference's walk applies
  // one after the other to the same signal: collect it (this node is its LAST stage) and hand it to the library as one
  // chain (groove_fx_chain_process: same bits as stage by stage, fewer passes over the block, and no copy from one
  // effect's block into the next one's).
  std::vector<groove_fx*> chain;
  Uid head = uid; // the chain's first stage: the one whose sources are summed
  if (fx->library_fx()) {
    chain.push_back(fx->library_fx());
    while (nodes_[head].sources.size() == 1) {
      const Uid s = nodes_[head].sources[0];
      Entity* se = nodes_[s].entity.get();
      if (!se || !se->is_effect() || fanout(s) != 1) break;
      Effect* sfx = static_cast<Effect*>(se);
      if (!sfx->library_fx() || sfx->lanes() != lanes) break;
      chain.push_back(sfx->library_fx());
      head = s;
    }
    std::reverse(chain.begin(), chain.end());
  }
  const std::vector<Uid>& sources = nodes_[head].sources;
  groove_block* io = nullptr;
  bool evaluated = false;
  if (!chain.empty() && sources.size() == 1 && fanout(sources[0]) == 1) {
    // one source that only this chain hears: lane for lane, its block is transformed where it lies
    groove_block* sb = nullptr;
    uint32_t sl = 0;
    if (eval(sources[0], frames, &sb, &sl)) return 1;
    evaluated = true;
    Entity* se = nodes_[sources[0]].entity.get();
    const bool in_place = sl == lanes && !(se->is_instrument() && !static_cast<Instrument*>(se)->output_is_scratch());
    if (sb && in_place) io = sb;
    else if (sb) {
      if (sl != lanes && lanes != 1) return fail("patch: source and sink lane counts differ");
      if (ensure_accum(nodes_[uid], lanes)) return 1;
      if (groove_block_accumulate(nodes_[uid].accum, sb, frames, 0)) return fail(groove_last_error(ctx_));
      io = nodes_[uid].accum;
    }
  }
  if (!io) {
    if (ensure_accum(nodes_[uid], lanes)) return 1;
    bool first = true;
    for (size_t i = 0; !evaluated && i < sources.size(); ++i) {
      groove_block* sb = nullptr;
      uint32_t sl = 0;
      if (eval(sources[i], frames, &sb, &sl)) return 1;
      if (!sb) continue;
      if (sl != lanes && lanes != 1) return fail("patch: source and sink lane counts differ");
      if (groove_block_accumulate(nodes_[uid].accum, sb, frames, first ? 0 : 1)) return fail(groove_last_error(ctx_));
      first = false;
    }
    if (first) { // an effect at the end of a chain with no input: silence in, so silence out
      if (groove_block_zero(nodes_[uid].accum)) return fail(groove_last_error(ctx_));
    }
    io = nodes_[uid].accum;
  }
  if (chain.empty()) {
    if (fx->transform_audio(io, frames)) return fail(groove_last_error(ctx_));
  } else if (groove_fx_chain_process(chain.data(), (uint32_t)chain.size(), io, frames)) return fail(groove_last_error(ctx_));
  *out_block = io;
  *out_lanes = lanes;
  return 0;
}
uint32_t Orchestrator::fanout(Uid uid) {
  if (!fanout_valid_) {
    fanout_.assign(nodes_.size(), 0);
    for (const Node& n : nodes_)
      for (Uid s : n.sources) fanout_[s]++;
    fanout_valid_ = true;
  }
  return uid < fanout_.size() ? fanout_[uid] : 0;
}
bool Orchestrator::direct_banks(std::vector<groove_bank*>& direct, std::vector<Uid>& rest) {
  direct.clear(); rest.clear();
  if (!fused_direct_ || ahead_eval_) return false; // (render-ahead walk: the instruments already hold their blocks)
  for (Uid s : nodes_[kMainMixerUid].sources) {
    Entity* e = nodes_[s].entity.get();
    groove_bank* b = (e && e->is_instrument() && fanout(s) == 1) ? static_cast<Instrument*>(e)->fused_bank() : nullptr;
    if (b && std::find(direct.begin(), direct.end(), b) == direct.end()) direct.push_back(b); else rest.push_back(s);
  }
  return !direct.empty();
}
int Orchestrator::gather_audio(uint32_t frames) {
  if (frames > bus_frames_) return fail("gather_audio: frames > block size");
  // Fast path: the main mixer is the identity, its output is the sum of its sources' lanes — so a bank
  // patched straight into it contributes its voices' sum and nothing else needs its block.  Everything else the mixer hears is
  // evaluated as before and summed onto the bus first; the direct banks then render fused onto it (one launch for several small
  // ones).  gather_audio's one sum over all sources, orchestrator.rs:397-410, in another order of additions.
  std::vector<groove_bank*> direct;
  std::vector<Uid> rest;
  if (direct_banks(direct, rest)) {
    int acc = 0;
    for (Uid u : rest) {
      groove_block* sb = nullptr;
      uint32_t sl = 0;
      if (eval(u, frames, &sb, &sl)) return 1;
      if (!sb) continue;
      groove_block* one[1] = {sb};
      if (groove_mix(ctx_, one, 1, frames, bus_, acc)) return fail(groove_last_error(ctx_));
      acc = 1;
    }
    if (groove_banks_render_mix_deferred(ctx_, direct.data(), (uint32_t)direct.size(), frames, bus_, acc)) return fail(groove_last_error(ctx_));
    return 0;
  }
  groove_block* b = nullptr;
  uint32_t lanes = 0;
  if (eval(kMainMixerUid, frames, &b, &lanes)) return 1;
  groove_block* arr[1] = {b};
  if (groove_mix(ctx_, arr, 1, frames, bus_, 0)) return fail(groove_last_error(ctx_));
  return 0;
}
uint64_t Orchestrator::performance_frames() const {
  uint64_t end = 0;
  for (auto& n : nodes_)
    if (n.entity && n.entity->is_controller())
      end = std::max(end, static_cast<Controller*>(n.entity.get())->end_units());
  const double beats = (double)end / MusicalTime::UNITS_IN_BEAT;
  return (uint64_t)std::ceil(beats * 60.0 / bpm_ * (double)sr_);
}
void Orchestrator::skip_to_start() { frames_ = 0; performing_ = true; ahead_primed_ = false; deferred_.clear(); }
int Orchestrator::control_effect(Uid target, uint32_t index, double value01) {
  Entity* e = get(target);
  if (e && e->is_instrument()) { // Controllable is generated for every entity (proc-macros/src/control.rs:171-183)
    // (never held back: in the render-ahead walk the block being sequenced is the one the instruments render NEXT — its predecessor's
    // render has been finished — while the effects are a block behind, which is what `deferring_` is for)
    if (static_cast<Instrument*>(e)->control_set_param(index, value01)) return fail(groove_last_error(ctx_));
    return 0;
  }
  if (deferring_) { deferred_.push_back({target, index, value01}); return 0; }
  if (e && e->is_effect()) return static_cast<Effect*>(e)->control_set_param(index, value01);
  return 0;
}
void Orchestrator::sequence_block(uint64_t at_frame, uint32_t frames) {
  // handle_work (orchestrator.rs:631-708): controllers see the block's musical-time range
  const uint64_t t0 = MusicalTime::frames_to_units(bpm_, sr_, at_frame);
  uint64_t t1 = MusicalTime::frames_to_units(bpm_, sr_, at_frame + frames);
  if (t1 == t0) t1 = t0 + 1;
  std::vector<MidiEvent> midi;
  for (auto& n : nodes_)
    if (n.entity && n.entity->is_controller()) static_cast<Controller*>(n.entity.get())->work(t0, t1, midi, *this);
  // broadcast_midi_messages (orchestrator.rs:710-754): every receiver on the channel
  for (const MidiEvent& m : midi) {
    auto range = midi_receivers_.equal_range(m.channel);
    for (auto it = range.first; it != range.second; ++it) {
      Instrument* ins = static_cast<Instrument*>(get(it->second));
      if (m.on) ins->note_on(m.key, m.velocity, at_frame); else ins->note_off(m.key, m.velocity, at_frame);
    }
  }
}
bool Orchestrator::ahead_instruments(std::vector<Instrument*>& out) {
  out.clear();
  std::vector<uint32_t> seen(nodes_.size(), 0);
  std::vector<Uid> stack{kMainMixerUid};
  while (!stack.empty()) {
    const Uid u = stack.back();
    stack.pop_back();
    if (++seen[u] > 1) return false; // heard through two paths: the block-by-block walk renders it twice; keep that
    Entity* e = nodes_[u].entity.get();
    if (!e) continue;
    if (e->is_instrument()) {
      Instrument* ins = static_cast<Instrument*>(e);
      if (!ins->supports_render_ahead()) return false;
      if (render_ahead_ < 2 && !ins->render_ahead_pays()) return false;
      out.push_back(ins);
    }
    for (Uid s : nodes_[u].sources) stack.push_back(s);
  }
  return true;
}
// One block of an offline run with the instruments one block ahead of the effects: the instruments'
// block b was rendered by the previous call (or by the priming below); this call sequences block b+1,
// starts its render on the side streams, and only then walks the effect graph of block b.  Effect
// automation computed while sequencing b+1 is held back until the effects have been given block b.
int Orchestrator::tick_ahead(StereoSample* out, uint32_t frames, uint32_t* ticks_completed, const std::vector<Instrument*>& instruments) {
  const uint64_t total = performance_frames();
  auto len_at = [&](uint64_t at) -> uint32_t { return at >= total ? 0u : (uint32_t)std::min<uint64_t>(frames, total - at); };
  const uint32_t done = len_at(frames_);
  if (done > 0) {
    if (!ahead_primed_) {
      sequence_block(frames_, done);
      for (Instrument* ins : instruments) if (ins->render_ahead(done)) return fail(groove_last_error(ctx_));
      ahead_primed_ = true;
    }
    for (Instrument* ins : instruments) if (ins->finish(done)) return fail(groove_last_error(ctx_));
    const uint32_t next = done == frames ? len_at(frames_ + done) : 0;
    if (next > 0) {
      deferring_ = true;
      sequence_block(frames_ + done, next);
      deferring_ = false;
      for (Instrument* ins : instruments) if (ins->render_ahead(next)) return fail(groove_last_error(ctx_));
    } else {
      ahead_primed_ = false;
    }
    ahead_eval_ = true;
    const int rc = gather_audio(done);
    ahead_eval_ = false;
    if (rc) return 1;
    if (out && groove_download(ctx_, bus_, &out[0].l, (size_t)done * 2)) return fail(groove_last_error(ctx_));
    for (const Deferred& d : deferred_) control_effect(d.target, d.index, d.value);
    deferred_.clear();
    frames_ += done; // clock.tick_batch(ticks_completed)
  }
  if (done < frames) performing_ = false;
  *ticks_completed = done;
  return 0;
}
int Orchestrator::tick_offline(StereoSample* out, uint32_t frames, uint32_t* ticks_completed) {
  std::vector<Instrument*> instruments;
  {
    std::vector<groove_bank*> direct;
    std::vector<Uid> rest;
    if (direct_banks(direct, rest) && rest.empty()) return tick(out, frames, ticks_completed); // nothing but fused banks: nothing to render ahead of
  }
  if (render_ahead_ && performing_ && ahead_instruments(instruments)) return tick_ahead(out, frames, ticks_completed, instruments);
  return tick(out, frames, ticks_completed);
}
int Orchestrator::tick(StereoSample* out, uint32_t frames, uint32_t* ticks_completed) {
  const uint64_t total = performance_frames();
  uint32_t done = frames;
  if (performing_) {
    if (frames_ >= total) done = 0;
    else if (frames_ + frames > total) done = (uint32_t)(total - frames_);
  }
  if (done > 0) {
    sequence_block(frames_, done);
    if (gather_audio(done)) return 1;
    if (out && groove_download(ctx_, bus_, &out[0].l, (size_t)done * 2)) return fail(groove_last_error(ctx_));
    if (performing_) frames_ += done; // clock.tick_batch(ticks_completed)
  }
  if (done < frames) performing_ = false;
  *ticks_completed = done;
  return 0;
}
int Orchestrator::run(uint32_t buffer_frames, std::vector<StereoSample>& out) {
  if (buffer_frames > GROOVE_BLOCK_FRAMES) return fail("run: buffer larger than the block size");
  skip_to_start();
  std::vector<StereoSample> buf(buffer_frames);
  for (;;) {
    uint32_t done = 0;
    if (tick_offline(buf.data(), buffer_frames, &done)) return 1;
    out.insert(out.end(), buf.begin(), buf.begin() + done);
    if (done < buffer_frames) break;
  }
  return 0;
}
int Orchestrator::run_performance(uint32_t buffer_frames, Performance& perf) {
  if (buffer_frames > GROOVE_BLOCK_FRAMES) return fail("run_performance: buffer larger than the block size");
  perf.sample_rate = sr_;
  skip_to_start();
  std::vector<StereoSample> buf(buffer_frames);
  for (;;) {
    uint32_t done = 0;
    if (tick_offline(buf.data(), buffer_frames, &done)) return 1;
    if (done < buffer_frames) break; // the final partial block is dropped (orchestrator.rs:827-836)
    perf.worker.insert(perf.worker.end(), buf.begin(), buf.end());
  }
  return 0;
}
int Orchestrator::send_performance_to_file(const Performance& perf, const std::string& path) {
  // hound::WavSpec{channels 2, sample_rate, 16-bit Int}; sample = (x * 32767) as i16 (helpers.rs:79-91)
  const size_t frames = perf.worker.size();
  std::vector<int16_t> pcm(frames * 2);
  float* dev = nullptr;
  if (frames) {
    if (groove_bus_create(ctx_, frames, &dev)) return fail(groove_last_error(ctx_));
    if (groove_upload(ctx_, dev, &perf.worker[0].l, frames * 2) || groove_bus_to_i16(ctx_, dev, frames, pcm.data())) {
      groove_bus_destroy(ctx_, dev);
      return fail(groove_last_error(ctx_));
    }
    groove_bus_destroy(ctx_, dev);
  }
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return fail("Couldn't create path from " + path);
  const uint32_t data_bytes = (uint32_t)(pcm.size() * 2), sr = perf.sample_rate;
  const uint32_t riff = 36 + data_bytes, fmt_len = 16, byte_rate = sr * 4;
  const uint16_t pcm_fmt = 1, ch = 2, align = 4, bits = 16;
  std::fwrite("RIFF", 1, 4, f); std::fwrite(&riff, 4, 1, f); std::fwrite("WAVEfmt ", 1, 8, f);
  std::fwrite(&fmt_len, 4, 1, f); std::fwrite(&pcm_fmt, 2, 1, f); std::fwrite(&ch, 2, 1, f);
  std::fwrite(&sr, 4, 1, f); std::fwrite(&byte_rate, 4, 1, f); std::fwrite(&align, 2, 1, f); std::fwrite(&bits, 2, 1, f);
  std::fwrite("data", 1, 4, f); std::fwrite(&data_bytes, 4, 1, f);
  if (!pcm.empty()) std::fwrite(pcm.data(), 2, pcm.size(), f);
  std::fclose(f);
  return 0;
}

} // namespace groove_host

// ====================================================================== C surface for tests / tools
using namespace groove_host;
extern "C" {
// BusStation C surface (tests restate the reference's unit test through it)
void* gh_bus_station_new() { return new groove_host::BusStation(); }
void gh_bus_station_free(void* h) { delete (groove_host::BusStation*)h; }
void gh_bus_station_add_send_route(void* h, uint32_t track, uint32_t aux, double amount) {
  ((groove_host::BusStation*)h)->add_send_route(track, groove_host::BusRoute{aux, amount});
}
void gh_bus_station_remove_send_route(void* h, uint32_t track, uint32_t aux) { ((groove_host::BusStation*)h)->remove_send_route(track, aux); }
void gh_bus_station_remove_track_sends(void* h, uint32_t track) { ((groove_host::BusStation*)h)->remove_track_sends(track); }
uint32_t gh_bus_station_tracks(void* h) { return (uint32_t)((groove_host::BusStation*)h)->tracks(); }
// number of sends of `track` (-1: the track has no list); the i-th route through aux_out / amount_out
int gh_bus_station_sends_for(void* h, uint32_t track) {
  const auto* r = ((groove_host::BusStation*)h)->sends_for(track);
  return r ? (int)r->size() : -1;
}
int gh_bus_station_send(void* h, uint32_t track, uint32_t i, uint32_t* aux_out, double* amount_out) {
  const auto* r = ((groove_host::BusStation*)h)->sends_for(track);
  if (!r || i >= r->size()) return 1;
  *aux_out = (*r)[i].aux_track_uid; *amount_out = (*r)[i].amount;
  return 0;
}
}
extern "C" {
void* gh_orchestrator_new(int device, uint32_t sample_rate, double bpm) {
  Orchestrator* o = new Orchestrator(device, sample_rate, bpm);
  if (!o->ctx()) { std::fprintf(stderr, "gh_orchestrator_new: %s\n", o->last_error().c_str()); delete o; return nullptr; }
  return o;
}
void gh_orchestrator_free(void* h) { delete (Orchestrator*)h; }
const char* gh_last_error(void* h) { return ((Orchestrator*)h)->last_error().c_str(); }
int gh_add_toy_source(void* h, double level) {
  Orchestrator* o = (Orchestrator*)h;
  return (int)o->add(std::unique_ptr<Entity>(new ToyAudioSource(o->ctx(), level)));
}
// One synth = one patch + `voices` voices (polyphony), summed to one lane.
int gh_add_welsh(void* h, const groove_welsh_params* patch, uint32_t voices) {
  Orchestrator* o = (Orchestrator*)h;
  std::vector<groove_welsh_params> p(voices, *patch);
  groove_bank* b = nullptr;
  if (groove_welsh_create(o->ctx(), p.data(), voices, &b)) { o->fail(groove_last_error(o->ctx())); return -1; }
  return (int)o->add(std::unique_ptr<Entity>(new VoiceBankInstrument(o->ctx(), b, voices, true, patch->amp_envelope.release, false)));
}
int gh_add_fm(void* h, const groove_fm_params* patch, uint32_t voices) {
  Orchestrator* o = (Orchestrator*)h;
  std::vector<groove_fm_params> p(voices, *patch);
  groove_bank* b = nullptr;
  if (groove_fm_create(o->ctx(), p.data(), voices, &b)) { o->fail(groove_last_error(o->ctx())); return -1; }
  return (int)o->add(std::unique_ptr<Entity>(new VoiceBankInstrument(o->ctx(), b, voices, true, patch->carrier_envelope.release, false)));
}
// Drumkit: one one-shot voice per sample buffer; MIDI key k plays buffer key_to_sample[k] (-1: none).
int gh_add_drumkit(void* h, const float* pcm, uint64_t frames, const groove_sample_desc* descs, uint32_t n_desc,
                   const int* key_to_sample /*[128]*/) {
  Orchestrator* o = (Orchestrator*)h;
  // voice v plays the sample mapped from key v, so that note_on(key) drives voice `key`
  std::vector<groove_sampler_params> p(128);
  std::vector<groove_sample_desc> d(descs, descs + n_desc);
  for (int k = 0; k < 128; ++k) {
    const int s = key_to_sample[k];
    p[k].sample_index = s >= 0 && (uint32_t)s < n_desc ? (uint32_t)s : 0;
    p[k].one_shot = 1;
    p[k].gain = s >= 0 ? 1.0f : 0.0f;
  }
  for (auto& x : d) x.root_hz = 0.0f; // drumkit: step 1 regardless of key
  groove_bank* b = nullptr;
  if (groove_sampler_create(o->ctx(), pcm, frames, d.data(), n_desc, p.data(), 128, &b)) { o->fail(groove_last_error(o->ctx())); return -1; }
  struct Drumkit : VoiceBankInstrument {
    using VoiceBankInstrument::VoiceBankInstrument;
    void note_on(uint8_t key, uint8_t vel, uint64_t) override { groove_note_event e{key, key, vel, 1, 0}; groove_bank_note_events(bank(), &e, 1); }
    void note_off(uint8_t, uint8_t, uint64_t) override {}
  };
  return (int)o->add(std::unique_ptr<Entity>(new Drumkit(o->ctx(), b, 128, true, 0.0, true)));
}
// Sampler (settings/src/instruments.rs:34-37, 81-88): one mono buffer, `voices` voices that step through it at note / root (SURVEY A.10:
// no interpolation), stopped by their note-off.
int gh_add_sampler(void* h, const float* pcm, uint64_t frames, double root_hz, uint32_t voices) {
  Orchestrator* o = (Orchestrator*)h;
  if (!pcm || !frames || !voices) { o->fail("gh_add_sampler: empty sample"); return -1; }
  groove_sample_desc d{0, (uint32_t)std::min<uint64_t>(frames, 0xFFFFFFFFu), (float)root_hz};
  std::vector<groove_sampler_params> p(voices);
  for (auto& x : p) { x.sample_index = 0; x.one_shot = 0; x.gain = 1.0f; }
  groove_bank* b = nullptr;
  if (groove_sampler_create(o->ctx(), pcm, frames, &d, 1, p.data(), voices, &b)) { o->fail(groove_last_error(o->ctx())); return -1; }
  return (int)o->add(std::unique_ptr<Entity>(new VoiceBankInstrument(o->ctx(), b, voices, true, 0.0, false)));
}
// ToyInstrument (settings/src/instruments.rs:27-28, 67-70; the reference's own test entity, source absent: docs/DSP_SPEC.md section 7): a
// sine oscillator at the note's pitch, gated by note-on / note-off, through a Dca — here an FM voice with index 0 and a gate for
// an envelope; `fake-value` is a dummy control of the reference's tests and has no sound.
int gh_add_toy_instrument(void* h, double dca_gain, double dca_pan) {
  Orchestrator* o = (Orchestrator*)h;
  groove_fm_params p{};
  p.ratio = 1.0; p.depth = 0.0f; p.beta = 0.0f;
  p.carrier_envelope = groove_envelope_params{0.0, 0.0, 1.0, 0.0};
  p.modulator_envelope = groove_envelope_params{0.0, 0.0, 1.0, 0.0};
  p.dca_gain = (float)dca_gain; p.dca_pan = (float)dca_pan;
  groove_bank* b = nullptr;
  if (groove_fm_create(o->ctx(), &p, 1, &b)) { o->fail(groove_last_error(o->ctx())); return -1; }
  return (int)o->add(std::unique_ptr<Entity>(new VoiceBankInstrument(o->ctx(), b, 1, true, 0.0, false)));
}
int gh_add_effect(void* h, uint32_t kind, const groove_fx_params* p) {
  Orchestrator* o = (Orchestrator*)h;
  return (int)o->add(std::unique_ptr<Entity>(new FxEffect(o->ctx(), kind, p, 1)));
}
int gh_patch(void* h, int source, int sink) { return ((Orchestrator*)h)->patch((Uid)source, (Uid)sink); }
int gh_patch_chain_to_main_mixer(void* h, const int* uids, uint32_t n) {
  std::vector<Uid> v(uids, uids + n);
  return ((Orchestrator*)h)->patch_chain_to_main_mixer(v);
}
void gh_unpatch_all(void* h) { ((Orchestrator*)h)->unpatch_all(); }
void gh_set_render_ahead(void* h, int mode) { ((Orchestrator*)h)->set_render_ahead(mode); }
void gh_set_fused_direct(void* h, int on) { ((Orchestrator*)h)->set_fused_direct(on != 0); }
int gh_connect_midi_downstream(void* h, int uid, int channel) { return ((Orchestrator*)h)->connect_midi_downstream((Uid)uid, (uint8_t)channel); }
int gh_add_timer(void* h, double beats) { return (int)((Orchestrator*)h)->add(std::unique_ptr<Entity>(new Timer(beats))); }
int gh_add_sequencer(void* h) { return (int)((Orchestrator*)h)->add(std::unique_ptr<Entity>(new Sequencer())); }
int gh_sequencer_insert(void* h, int uid, int channel, int key, double start_beat, double duration_beats) {
  Entity* e = ((Orchestrator*)h)->get((Uid)uid);
  if (!e || !e->is_controller()) return 1;
  static_cast<Sequencer*>(e)->insert((uint8_t)channel, (uint8_t)key, start_beat, duration_beats);
  return 0;
}
int gh_sequencer_set_end(void* h, int uid, double beats) {
  Entity* e = ((Orchestrator*)h)->get((Uid)uid);
  if (!e || !e->is_controller()) return 1;
  static_cast<Sequencer*>(e)->set_end_beats(beats);
  return 0;
}
int gh_add_control_trip(void* h, int target_uid, const char* param_name, double start_beat) {
  Orchestrator* o = (Orchestrator*)h;
  Entity* e = o->get((Uid)target_uid);
  if (!e || !(e->is_effect() || e->is_instrument())) { o->fail("control trip target is neither an effect nor an instrument"); return -1; }
  const int idx = e->is_effect() ? static_cast<Effect*>(e)->control_index_for_name(param_name) : static_cast<Instrument*>(e)->control_index_for_name(param_name);
  if (idx < 0) { o->fail(std::string("unknown control name ") + param_name); return -1; }
  return (int)o->add(std::unique_ptr<Entity>(new ControlTrip((Uid)target_uid, (uint32_t)idx, start_beat)));
}
int gh_control_trip_add_step(void* h, int uid, int kind, double start, double end, double beats) {
  Entity* e = ((Orchestrator*)h)->get((Uid)uid);
  if (!e || !e->is_controller()) return 1;
  ControlStep s; s.kind = (ControlStep::Kind)kind; s.start = start; s.end = end; s.beats = beats;
  static_cast<ControlTrip*>(e)->add_step(s);
  return 0;
}
double gh_control_step_value(int kind, double start, double end, double t01) {
  ControlStep s; s.kind = (ControlStep::Kind)kind; s.start = start; s.end = end;
  return ControlTrip::value_at(s, t01);
}
int gh_last_allocated_voice(void* h, int uid) {
  Entity* e = ((Orchestrator*)h)->get((Uid)uid);
  if (!e || !e->is_instrument()) return -1;
  return (int)static_cast<VoiceBankInstrument*>(e)->last_allocated_voice();
}
int gh_gather_audio(void* h, uint32_t frames, float* out_interleaved) {
  Orchestrator* o = (Orchestrator*)h;
  uint32_t done = 0;
  return o->tick((StereoSample*)out_interleaved, frames, &done);
}
uint64_t gh_performance_frames(void* h) { return ((Orchestrator*)h)->performance_frames(); }
// Runs the whole performance; returns the number of frames written (<= cap), or -1.
int64_t gh_run(void* h, uint32_t buffer_frames, float* out_interleaved, uint64_t cap_frames, int performance_mode) {
  Orchestrator* o = (Orchestrator*)h;
  std::vector<StereoSample> out;
  if (performance_mode) {
    Performance p;
    if (o->run_performance(buffer_frames, p)) return -1;
    out.swap(p.worker);
  } else if (o->run(buffer_frames, out)) return -1;
  const uint64_t n = std::min<uint64_t>(cap_frames, out.size());
  if (n) std::memcpy(out_interleaved, out.data(), n * sizeof(StereoSample));
  return (int64_t)out.size();
}
int gh_render_to_wav(void* h, uint32_t buffer_frames, const char* path) {
  Orchestrator* o = (Orchestrator*)h;
  Performance p;
  if (o->run_performance(buffer_frames, p)) return 1;
  return o->send_performance_to_file(p, path);
}
}
