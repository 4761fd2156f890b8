// project.hpp — project (JSON / JSON5) and Welsh patch (JSON) loading for the compiled host layer.
//
// Mirrors the reference's `settings` crate:
//   SongSettings                      /root/reference/settings/src/songs.rs:19-56, 91-306
//   DeviceSettings / *Settings enums  settings/src/lib.rs:40-46, instruments.rs:26-39, effects.rs:19-56
//   PatternSettings / TrackSettings   settings/src/lib.rs:48-88
//   ControlPath / ControlTrip         settings/src/controllers.rs:18-99
//   WelshPatchSettings                settings/src/patches.rs:20-47, 87-170, 204-314
// Parsing is split from instantiation so that the schema handling can be tested without a GPU.
#pragma once
#include "groove_host.hpp"
#include "json5.hpp"
#include <stdexcept>

namespace groove_host {

struct ProjectDesc {
  struct Device {
    std::string id;
    std::string kind;        // "welsh", "welsh-raw", "drumkit", "sampler", "fm-synthesizer", "toy-instrument", or an effect name
    bool is_effect = false;
    int midi_in = 0;
    std::string name;        // welsh patch name / drumkit name / sampler filename
    double root = 0.0;       // sampler root frequency (0: the file's own — its smpl / acid chunk — else 440 Hz)
    double toy_value = 0.0;  // toy-instrument: `fake-value` (a dummy control of the reference's tests)
    double dca_gain = 1.0, dca_pan = 0.0; // toy-instrument: its Dca
    groove_welsh_params welsh{};
    groove_fm_params fm{};
    uint32_t fx_kind = GROOVE_FX_MIXER;
    groove_fx_params fx{};
  };
  struct Note { int channel, key; double start_beat, duration_beats; };
  struct Trip { std::string id, target, param; double start_beat = 0.0; std::vector<ControlStep> steps; };

  std::string title;
  std::string project_dir;   // directory of the project file (sample files are looked for there too); empty for parse_project(text)
  double bpm = 128.0;
  int ts_top = 4, ts_bottom = 4;
  std::vector<Device> devices;
  std::vector<std::vector<std::string>> patch_cables;
  std::vector<Note> notes;   // tracks x patterns flattened to absolute beats
  double end_beats = 0.0;    // the sequencer ends at the end of its last full measure
  std::vector<Trip> trips;
  std::vector<std::string> warnings; // the reference eprintln!s and continues (songs.rs:137, 152-156)
};

// WelshPatchSettings::derive_welsh_synth_params (settings/src/patches.rs:87-170).
groove_welsh_params welsh_params_from_patch_json(const json5::Value& patch, std::vector<std::string>* warnings);
// WelshPatchSettings::patch_name_to_settings_name (patches.rs:51-55): "ElectricPiano" → "electric-piano".
std::string patch_name_to_settings_name(const std::string& name);

// SongSettings::new_from_project_file + the parsing half of instantiate().  `assets_root` is the
// directory that holds patches/welsh/*.json and samples/ (the reference's assets/); Welsh devices
// need it, everything else parses without it.
ProjectDesc parse_project(const std::string& json_text, const std::string& assets_root);
ProjectDesc parse_project_file(const std::string& path, const std::string& assets_root);
// A compact JSON rendering of a ProjectDesc (used by the tests).
std::string describe(const ProjectDesc& p);

// WelshSynthParams as the project file carries it for `welsh-raw` (settings/src/instruments.rs:30-31; the struct settings/src/patches.rs:110-169
// builds): {"voice": {"oscillator-1": {"waveform", "frequency-tune"}, .., "amp-envelope", "lfo", "lfo-routing", "lfo-depth", "filter":
// {"cutoff", "passband-ripple"}, "filter-cutoff-start", "filter-cutoff-end", "filter-envelope", "dca"}, "dca"}; envelope times in seconds.
groove_welsh_params welsh_params_from_raw_json(const json5::Value& params, std::vector<std::string>* warnings);
// The MIDI root note a WAV file carries: the `smpl` chunk's unity note, else the `acid` chunk's root note (when its flags say it is
// set; test-data/samples/riff-acidized.wav: 57), else -1.
int read_wav_root_note(const std::string& path);
// Mono PCM from a WAV file (16/24/32-bit int or 32-bit float, any channel count: channels averaged).
bool read_wav_mono(const std::string& path, std::vector<float>& out, uint32_t* sample_rate, std::string* err);

// The instantiate() half: devices → entities, patch cables, MIDI routing, tracks → Sequencer,
// trips → ControlTrip.  With `synthetic_kit` the drumkit uses a generated sample bank instead of
// <assets>/samples/elphnt.io/707 (the GPU box has no reference assets).
int instantiate(Orchestrator& o, const ProjectDesc& p, const std::string& assets_root, bool synthetic_kit);

} // namespace groove_host
