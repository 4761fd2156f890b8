// groove_host.hpp — compiled host layer above the C ABI (include/groove_hip.h).
//
// The reference's host code is Rust; no Rust toolchain exists in the build image, so the
// host side that sits above the FFI is written in C++ and mirrors the reference's
// operator surface for this path, name for name:
//
//   Orchestrator::{add, patch, patch_chain_to_main_mixer, unpatch_all,
//                  connect_midi_downstream, tick, gather_audio, run, run_performance,
//                  update_sample_rate}      /root/reference/orchestration/src/orchestrator.rs
//                                           :136-142, 263-325, 472-490, 367-470, 788-877
//   Performance{sample_rate, worker}        orchestrator.rs:32-46
//   IsInstrument / IsEffect / IsController  proc-macros/src/entity.rs:29-139
//   IOHelper::send_performance_to_file      orchestration/src/helpers.rs:74-97
//
// Block semantics.  The reference evaluates the patch graph once per FRAME
// (gather_audio, orchestrator.rs:367-470); here the same post-order traversal runs once
// per BLOCK on device blocks, which is equivalent because entities never reference each
// other and events are block-granular already (tick(): handle_work once, then gather).
// A node's output is a device block with `lanes` lanes.  A WelshSynth with 8 voices is
// one entity whose voices are summed to ONE lane (the reference's Synthesizer sums its
// voice store); "batched" instruments (`sum_voices = false`) keep one lane per voice so
// that per-voice effect chains (BASELINE config #3) run as wide effect banks.  An effect
// sums ALL its sources, then transforms the sum once (orchestrator.rs:438-457, pinned by
// the fan-in test :1642-1668); sources with more lanes than the sink are lane-summed.
#pragma once
#include "../../include/groove_hip.h"
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <map>

namespace groove_host {

using Uid = size_t;
constexpr Uid kMainMixerUid = 0; // MAIN_MIXER_UVID, orchestrator.rs:104, 543-546

struct StereoSample { float l, r; };

// Performance, orchestrator.rs:32-46 (the FIFO worker is a plain vector here).
struct Performance {
  uint32_t sample_rate = GROOVE_DEFAULT_SAMPLE_RATE;
  std::vector<StereoSample> worker;
};

// MusicalTime: 65,536 units per beat (src/mini/transport.rs:157-176, doc/designs/time.md:94-98).
struct MusicalTime {
  static constexpr uint64_t UNITS_IN_BEAT = 65536;
  uint64_t units = 0;
  static MusicalTime from_beats(double beats) { return {(uint64_t)(beats * UNITS_IN_BEAT + 0.5)}; }
  static uint64_t frames_to_units(double bpm, uint32_t sr, uint64_t frames) {
    return (uint64_t)((double)frames * bpm / 60.0 / (double)sr * (double)UNITS_IN_BEAT);
  }
  double beats() const { return (double)units / UNITS_IN_BEAT; }
};

class Orchestrator;

class Entity {
 public:
  virtual ~Entity() = default;
  Uid uid = 0;
  std::string name;
  virtual bool is_instrument() const { return false; }
  virtual bool is_effect() const { return false; }
  virtual bool is_controller() const { return false; }
};

// IsInstrument: Generates<StereoSample> + Ticks + HandlesMidi.
class Instrument : public Entity {
 public:
  bool is_instrument() const override { return true; }
  virtual uint32_t lanes() const = 0;          // lanes of the block it outputs
  virtual groove_block* output() = 0;
  virtual int tick(uint32_t frames) = 0;       // render `frames` into output()
  // Does every tick / finish write the whole of output() anew?  Then a chain that is the block's only listener may
  // transform it where it lies; a source that keeps a constant block (ToyAudioSource) must be copied first.
  virtual bool output_is_scratch() const { return true; }
  // HandlesMidi::handle_midi_message (note on/off only on this path)
  virtual void note_on(uint8_t key, uint8_t velocity, uint64_t now_frame) = 0;
  virtual void note_off(uint8_t key, uint8_t velocity, uint64_t now_frame) = 0;
  // Render-ahead protocol of the offline runs (Orchestrator::run / run_performance): render_ahead()
  // starts the NEXT block's render beside whatever the ctx stream does (groove_bank_render_async)
  // without disturbing output(); finish() makes that block the current output().
  virtual bool supports_render_ahead() const { return false; }
  virtual bool render_ahead_pays() const { return true; }
  virtual int render_ahead(uint32_t frames) { (void)frames; return 0; }
  virtual int finish(uint32_t frames) { return tick(frames); }
  // The library bank behind the instrument, if all it does is render that bank: patched STRAIGHT into the main mixer and heard by
  // nothing else, such an instrument needs no voice block at all — the orchestrator renders it fused onto the bus
  // (Orchestrator::gather_audio's fast path).
  virtual groove_bank* fused_bank() { return nullptr; }
  // Controllable (proc-macros/src/control.rs:171-249: generated for EVERY entity, instruments included): a ControlTrip whose target
  // is an instrument drives these (entities/src/controllers/control_trip.rs:184-254).  -1 / 1: no such control.
  virtual int control_index_for_name(const std::string&) const { return -1; }
  virtual int control_set_param(uint32_t index, double value01) { (void)index; (void)value01; return 1; }
};

// IsEffect: TransformsAudio.
class Effect : public Entity {
 public:
  bool is_effect() const override { return true; }
  virtual uint32_t lanes() const = 0;
  virtual int transform_audio(groove_block* inout, uint32_t frames) = 0;
  virtual groove_fx* library_fx() { return nullptr; } // the library effect behind it, if it is one (chains of those are fused)
  virtual int control_set_param(uint32_t index, double value01) { (void)index; (void)value01; return 1; }
  virtual int control_index_for_name(const std::string&) const { return -1; }
};

// IsController: Controls (update_time / work / is_finished), src/mini/transport.rs:116-151.
struct MidiEvent { uint8_t channel, key, velocity; bool on; };
class Controller : public Entity {
 public:
  bool is_controller() const override { return true; }
  // events that fall inside [start, end) musical-time units, appended to `out`
  virtual void work(uint64_t start_units, uint64_t end_units, std::vector<MidiEvent>& out,
                    Orchestrator& o) = 0;
  virtual bool is_finished(uint64_t end_units) const = 0;
  virtual uint64_t end_units() const = 0;
  virtual void skip_to_start() {}
};

// ---- concrete entities -------------------------------------------------------------------
// A synth = ONE patch + a voice store (first-idle allocation, A.6); voices summed to 1 lane.
class VoiceBankInstrument : public Instrument {
 public:
  VoiceBankInstrument(groove_ctx* ctx, groove_bank* bank, uint32_t voices, bool sum_voices,
                      double release_seconds, bool one_voice_per_key);
  ~VoiceBankInstrument() override;
  uint32_t lanes() const override { return sum_voices_ ? 1 : voices_; }
  groove_block* output() override { return sum_voices_ ? summed_ : block_; }
  int tick(uint32_t frames) override;
  bool supports_render_ahead() const override { return true; }
  // A bank the library renders time-parallel (one wavefront per voice: ~20 us per block) is shorter than what the
  // side-stream hand-over costs (event packets, a cross-queue wait): render-ahead only pays for the serial forms.
  bool render_ahead_pays() const override { return !short_render_; }
  int render_ahead(uint32_t frames) override;
  int finish(uint32_t frames) override;
  void note_on(uint8_t key, uint8_t velocity, uint64_t now_frame) override;
  void note_off(uint8_t key, uint8_t velocity, uint64_t now_frame) override;
  groove_bank* bank() { return bank_; }
  groove_bank* fused_bank() override { return bank_; }
  // the controls the library's banks expose (include/groove_types.h GROOVE_CTL_WELSH_*; Welsh banks only), for every voice of the synth
  int control_index_for_name(const std::string& name) const override;
  int control_set_param(uint32_t index, double value01) override;
  uint32_t last_allocated_voice() const { return last_voice_; }
 private:
  groove_ctx* ctx_;
  groove_bank* bank_;
  uint32_t voices_;
  bool sum_voices_;
  double release_seconds_;
  bool per_key_; // Drumkit: one voice per note (A.10)
  bool short_render_ = false;
  groove_block* block_ = nullptr;
  groove_block* block_next_ = nullptr; // render-ahead: the block being rendered while block_ is consumed
  groove_block* summed_ = nullptr;
  std::vector<int> key_of_voice_;        // -1 = free
  std::vector<uint64_t> busy_until_;     // frame until which the voice may still sound
  std::vector<uint64_t> started_;
  uint32_t last_voice_ = 0;
};

// ToyAudioSource{level}: constant-level instrument used by the reference's mix-bus tests.
class ToyAudioSource : public Instrument {
 public:
  ToyAudioSource(groove_ctx* ctx, double level);
  ~ToyAudioSource() override;
  uint32_t lanes() const override { return 1; }
  groove_block* output() override { return block_; }
  int tick(uint32_t frames) override;
  bool supports_render_ahead() const override { return true; } // constant block: nothing to render
  bool output_is_scratch() const override { return false; }
  void note_on(uint8_t, uint8_t, uint64_t) override {}
  void note_off(uint8_t, uint8_t, uint64_t) override {}
 private:
  groove_ctx* ctx_;
  float level_;
  groove_block* block_ = nullptr;
  std::vector<float> host_;
};

class FxEffect : public Effect {
 public:
  FxEffect(groove_ctx* ctx, uint32_t kind, const groove_fx_params* p, uint32_t lanes);
  ~FxEffect() override;
  uint32_t lanes() const override { return lanes_; }
  int transform_audio(groove_block* inout, uint32_t frames) override;
  groove_fx* library_fx() override { return fx_; }
  int control_set_param(uint32_t index, double value01) override;
  int control_index_for_name(const std::string& name) const override;
 private:
  groove_ctx* ctx_;
  groove_fx* fx_ = nullptr;
  uint32_t lanes_;
};

// Timer: finishes after N beats (groove-toys Timer, orchestrator.rs:1408-1415).
class Timer : public Controller {
 public:
  explicit Timer(double beats) : end_(MusicalTime::from_beats(beats).units) {}
  void work(uint64_t, uint64_t, std::vector<MidiEvent>&, Orchestrator&) override {}
  bool is_finished(uint64_t end_units) const override { return end_units >= end_; }
  uint64_t end_units() const override { return end_; }
 private:
  uint64_t end_;
};

// Sequencer: notes at musical-time positions on a MIDI channel (settings/src/songs.rs:210-249
// inserts pattern notes; the beat sequencer finishes at the end of its last full measure).
class Sequencer : public Controller {
 public:
  void insert(uint8_t channel, uint8_t key, double start_beat, double duration_beats);
  void set_end_beats(double beats) { end_ = MusicalTime::from_beats(beats).units; explicit_end_ = true; }
  void work(uint64_t start_units, uint64_t end_units, std::vector<MidiEvent>& out, Orchestrator&) override;
  bool is_finished(uint64_t end_units) const override { return end_units >= end_; }
  uint64_t end_units() const override { return end_; }
 private:
  struct Ev { uint64_t at; uint8_t channel, key; bool on; };
  std::vector<Ev> events_;
  uint64_t end_ = 0;
  bool explicit_end_ = false;
};

// ControlTrip: automation of one Controllable parameter along a path of steps
// (entities/src/controllers/control_trip.rs:7-26, 99-142, 257-264).
struct ControlStep {
  enum Kind { FLAT, SLOPE, LOGARITHMIC, EXPONENTIAL, TRIGGERED } kind = FLAT;
  double start = 0.0, end = 0.0; // FLAT uses start as its value
  double beats = 1.0;            // duration of this step
};
class ControlTrip : public Controller {
 public:
  ControlTrip(Uid target, uint32_t control_index, double start_beat) : target_(target), index_(control_index), start_(start_beat) {}
  void add_step(const ControlStep& s) { steps_.push_back(s); }
  void work(uint64_t start_units, uint64_t end_units, std::vector<MidiEvent>& out, Orchestrator& o) override;
  bool is_finished(uint64_t end_units) const override { return end_units >= end_units_(); }
  uint64_t end_units() const override { return end_units_(); }
  static double value_at(const ControlStep& s, double t01);
 private:
  uint64_t end_units_() const;
  Uid target_;
  uint32_t index_;
  double start_;
  std::vector<ControlStep> steps_;
  double last_sent_ = -1.0;
};

// ---- the orchestrator ----------------------------------------------------------------------
class Orchestrator {
 public:
  Orchestrator(int device, uint32_t sample_rate, double bpm);
  ~Orchestrator();
  Orchestrator(const Orchestrator&) = delete;

  groove_ctx* ctx() { return ctx_; }
  const std::string& last_error() const { return err_; }
  uint32_t sample_rate() const { return sr_; }
  double bpm() const { return bpm_; }
  void set_bpm(double bpm) { bpm_ = bpm; } // Clock::set_bpm (orchestrator.rs:888-890)
  int update_sample_rate(uint32_t hz);

  // Orchestrator::add (orchestrator.rs:136-142): takes ownership, returns the Uid.
  Uid add(std::unique_ptr<Entity> e);
  Entity* get(Uid uid);
  // Orchestrator::patch (orchestrator.rs:263-304): source output → sink input.  0 on success.
  int patch(Uid source, Uid sink);
  int patch_chain_to_main_mixer(const std::vector<Uid>& uids);
  void unpatch_all();
  // Orchestrator::connect_midi_downstream (orchestrator.rs:472-490)
  int connect_midi_downstream(Uid receiver, uint8_t channel);

  // Orchestrator::tick (orchestrator.rs:856-877): events for this block, gather_audio, clock
  // advance.  Writes frames into `out` and returns ticks_completed (< frames ends the run).
  int tick(StereoSample* out, uint32_t frames, uint32_t* ticks_completed);
  // Orchestrator::gather_audio (orchestrator.rs:367-470) for one block into the device bus.
  int gather_audio(uint32_t frames);
  // Orchestrator::run / run_performance (orchestrator.rs:788-846).  `run` keeps the final
  // partial block, `run_performance` drops it (quirk, :827-836).
  int run(uint32_t buffer_frames, std::vector<StereoSample>& out);
  int run_performance(uint32_t buffer_frames, Performance& perf);
  void skip_to_start();
  // Offline runs render the instruments of block b+1 on the library's side streams while the effect
  // graph of block b runs (same samples; DESIGN.md "Render-ahead").  On by default; off = block by block.
  // 0 = block by block, 1 = where it pays (default), 2 = whenever the graph allows it (tests).
  void set_render_ahead(int mode) { render_ahead_ = mode; }
  // Instruments patched straight into the main mixer (and heard by nothing else) render FUSED onto the bus — no voice block, one
  // launch for all of them when they are small (groove_banks_render_mix_deferred) — instead of block by block through the mixer's
  // accumulator.  Same sum to fp32 rounding.  On by default; off = the entity-boundary walk for every source (tests, A/B).
  void set_fused_direct(bool on) { fused_direct_ = on; }
  // ControlTrip -> effect parameter.  While the controllers are run one block ahead of the effects
  // the update is held back until the effects have processed the current block.
  int control_effect(Uid target, uint32_t index, double value01);
  uint64_t clock_frames() const { return frames_; }
  uint64_t performance_frames() const; // ceil(end of the last controller)

  // IOHelper::send_performance_to_file (helpers.rs:74-97): 16-bit stereo PCM WAV, quantised on the device.
  int send_performance_to_file(const Performance& perf, const std::string& path);

  int fail(const std::string& msg) { err_ = msg; return 1; }

 private:
  struct Node {
    std::unique_ptr<Entity> entity;
    std::vector<Uid> sources;          // audio_sink_uid_to_source_uids
    groove_block* accum = nullptr;     // effect input sum / output block
    uint32_t accum_lanes = 0;
  };
  int eval(Uid uid, uint32_t frames, groove_block** out_block, uint32_t* out_lanes);
  int ensure_accum(Node& n, uint32_t lanes);
  uint32_t fanout(Uid uid); // how many sinks hear this node
  // handle_work + broadcast_midi_messages for the block [at_frame, at_frame + frames)
  void sequence_block(uint64_t at_frame, uint32_t frames);
  // the instruments the main mixer hears, if every one of them is heard exactly once and can render ahead
  bool ahead_instruments(std::vector<Instrument*>& out);
  // the main mixer's sources split into banks that can render fused onto the bus and the rest (chains, toy sources, shared instruments)
  bool direct_banks(std::vector<groove_bank*>& direct, std::vector<Uid>& rest);
  int tick_ahead(StereoSample* out, uint32_t frames, uint32_t* ticks_completed, const std::vector<Instrument*>& instruments);
  int tick_offline(StereoSample* out, uint32_t frames, uint32_t* ticks_completed);

  groove_ctx* ctx_ = nullptr;
  uint32_t sr_;
  double bpm_;
  std::string err_;
  std::vector<Node> nodes_;
  std::vector<uint32_t> fanout_; // per node, rebuilt after a patch change
  bool fanout_valid_ = false;
  std::multimap<uint8_t, Uid> midi_receivers_;
  float* bus_ = nullptr; // device [block][2]
  uint32_t bus_frames_ = 0;
  uint64_t frames_ = 0;
  bool performing_ = false;
  int render_ahead_ = 1;
  bool fused_direct_ = true;
  bool ahead_primed_ = false;   // the current block's instruments were rendered by the previous tick_ahead
  bool ahead_eval_ = false;     // eval(): instruments already hold their block
  bool deferring_ = false;      // controllers are being run for the NEXT block
  struct Deferred { Uid target; uint32_t index; double value; };
  std::vector<Deferred> deferred_;
};

// BusStation (src/mini/bus_station.rs:7-52): which track sends how much of its signal to which aux
// track.  At the reference commit it is a routing table only (nothing renders through it yet), so
// this is the table, with the behaviour of its code and unit test (:55-140): adding a route appends
// (the test's "should replace the prior one" only counts tracks; the code pushes, so does this),
// removing a route drops every route to that aux and leaves the track's (possibly empty) list in
// place, removing a track's sends leaves an empty list for it.
struct BusRoute { uint32_t aux_track_uid; double amount; };
class BusStation {
 public:
  void add_send_route(uint32_t track_uid, const BusRoute& route) {
    send_routes_[track_uid].push_back(route);
  }
  void remove_send_route(uint32_t track_uid, uint32_t aux_track_uid) {
    auto it = send_routes_.find(track_uid);
    if (it == send_routes_.end()) return;
    std::vector<BusRoute>& routes = it->second;
    for (size_t i = 0; i < routes.size();)
      if (routes[i].aux_track_uid == aux_track_uid) routes.erase(routes.begin() + (long)i); else ++i;
  }
  void remove_track_sends(uint32_t track_uid) { send_routes_[track_uid].clear(); }
  const std::vector<BusRoute>* sends_for(uint32_t track_uid) const {
    auto it = send_routes_.find(track_uid);
    return it == send_routes_.end() ? nullptr : &it->second;
  }
  size_t tracks() const { return send_routes_.size(); }
 private:
  std::map<uint32_t, std::vector<BusRoute>> send_routes_;
};

} // namespace groove_host
