/* groove_types.h — plain-old-data parameter structs shared by the C ABI
 * (include/groove_hip.h), the CPU oracle (oracle/) and the C++ host layer.
 *
 * Every struct mirrors a `*Params` struct the reference constructs its entities
 * from (`#[derive(Params)]` → `FooParams` + `Foo::new_with(&FooParams)`,
 * /root/reference/proc-macros/src/params.rs:110-139).  Field meaning and units
 * follow docs/DSP_SPEC.md.  All citations are relative to /root/reference.
 */
#ifndef GROOVE_TYPES_H
#define GROOVE_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GROOVE_DEFAULT_SAMPLE_RATE 44100u /* SampleRate::DEFAULT, src/bin/groove-cli.rs:115-119 */
#define GROOVE_BLOCK_FRAMES 256u          /* BASELINE.json configs: 256-frame blocks        */
#define GROOVE_ALL_VOICES 0xFFFFFFFFu

/* WaveformType, settings/src/patches.rs:173-189 (serde kebab-case names in comments). */
typedef enum {
  GROOVE_WAVE_NONE = 0,          /* "none"          → 0.0                      */
  GROOVE_WAVE_SINE = 1,          /* "sine"                                      */
  GROOVE_WAVE_SQUARE = 2,        /* "square"        (pulse, duty 0.5)           */
  GROOVE_WAVE_PULSE_WIDTH = 3,   /* "pulse-width"(f32 duty)                     */
  GROOVE_WAVE_TRIANGLE = 4,      /* "triangle"                                  */
  GROOVE_WAVE_SAWTOOTH = 5,      /* "sawtooth"                                  */
  GROOVE_WAVE_NOISE = 6,         /* "noise"         (u32 generator, bit-exact)  */
  GROOVE_WAVE_DEBUG_ZERO = 7,    /* "debug-zero"    → 0.0                       */
  GROOVE_WAVE_DEBUG_MAX = 8,     /* "debug-max"     → +1.0                      */
  GROOVE_WAVE_DEBUG_MIN = 9,     /* "debug-min"     → -1.0                      */
  GROOVE_WAVE_TRIANGLE_SINE = 10 /* "triangle-sine"                             */
} groove_waveform;

/* LfoRoutingType, settings/src/patches.rs:269-290 (values 0-4), plus the routings the shipped Welsh
 * patch files carry that the enum does not know yet (the files under assets/patches/welsh, field "routing": "pitch-osc2",
 * "pw-osc1", "pw-osc2", "resonance", "cutoff-amp").  docs/DSP_SPEC.md §6 defines them. */
typedef enum {
  GROOVE_LFO_NONE = 0,
  GROOVE_LFO_AMPLITUDE = 1,
  GROOVE_LFO_PITCH = 2,
  GROOVE_LFO_PULSE_WIDTH = 3,
  GROOVE_LFO_FILTER_CUTOFF = 4,
  GROOVE_LFO_PITCH_OSC2 = 5,  /* "pitch-osc2": frequency modulation of oscillator 2 only          */
  GROOVE_LFO_PW_OSC1 = 6,     /* "pw-osc1":    pulse width of oscillator 1 only                   */
  GROOVE_LFO_PW_OSC2 = 7,     /* "pw-osc2":    pulse width of oscillator 2 only                   */
  GROOVE_LFO_RESONANCE = 8,   /* "resonance":  the 24 dB filter's passband ripple, every frame    */
  GROOVE_LFO_CUTOFF_AMP = 9   /* "cutoff-amp": filter cutoff and amplitude together               */
} groove_lfo_routing;
#define GROOVE_LFO_ROUTING_COUNT 10
#define GROOVE_WAVEFORM_COUNT 11

/* EnvelopeParams{attack, decay, sustain, release}, settings/src/patches.rs:133-138.
 * Times are SECONDS (the unit the shipped patch JSON carries, e.g.
 * assets/patches/welsh/cello.json:45-56); 0 = instant. sustain is a Normal 0..1. */
typedef struct {
  double attack;
  double decay;
  double sustain;
  double release;
} groove_envelope_params;

/* OscillatorParams{waveform, frequency, frequency_tune} + fixed frequency,
 * settings/src/patches.rs:112-121, 93-100, 260-266. */
typedef struct {
  uint32_t waveform; /* groove_waveform */
  float duty;        /* PulseWidth(f32) duty cycle; ignored otherwise */
  double tune;       /* Ratio; OscillatorTune → Ratio, patches.rs:209-219 */
  double fixed_hz;   /* > 0: set_fixed_frequency(note_to_frequency(n)), patches.rs:94-100 */
} groove_oscillator_params;

/* WelshVoiceParams, settings/src/patches.rs:110-164. */
typedef struct {
  groove_oscillator_params oscillator_1;
  groove_oscillator_params oscillator_2;
  uint32_t oscillator_2_sync;
  float oscillator_mix; /* Normal: weight of oscillator 1, patches.rs:123-132 */
  groove_envelope_params amp_envelope;
  groove_envelope_params filter_envelope;
  uint32_t lfo_waveform;
  uint32_t lfo_routing; /* groove_lfo_routing */
  double lfo_frequency; /* Hz */
  float lfo_depth;      /* Normal, patches.rs:304-314 */
  float filter_cutoff_hz;       /* BiQuadFilterLowPass24dbParams.cutoff, patches.rs:146-147 */
  float filter_passband_ripple; /* ….passband_ripple = denormalize_q(resonance), :148 */
  float filter_cutoff_start;    /* frequency_to_percent(12 dB cutoff), :150-152 */
  float filter_cutoff_end;      /* filter_envelope_weight, :153 */
  float dca_gain; /* DcaParams.gain, :160-163 */
  float dca_pan;  /* DcaParams.pan (BipolarNormal) */
} groove_welsh_params;

/* FmSynthParams{ratio, depth, beta, carrier_envelope, modulator_envelope, dca},
 * settings/src/patches.rs:691-715. */
typedef struct {
  double ratio;
  float depth;
  float beta;
  groove_envelope_params carrier_envelope;
  groove_envelope_params modulator_envelope;
  float dca_gain;
  float dca_pan;
} groove_fm_params;

/* One buffer of the shared mono sample bank (Sampler/Drumkit,
 * settings/src/instruments.rs:34-37, 81-88). */
typedef struct {
  uint64_t offset; /* first frame of this buffer inside bank_pcm */
  uint32_t length; /* frames */
  float root_hz;   /* SamplerParams.root; <= 0 ⇒ drumkit (step 1.0 regardless of key) */
} groove_sample_desc;

/* Per-voice sampler assignment. */
typedef struct {
  uint32_t sample_index; /* which groove_sample_desc */
  uint32_t one_shot;     /* 1: note-off ignored (Drumkit); 0: note-off stops (Sampler) */
  float gain;            /* linear gain applied to the fetched sample */
} groove_sampler_params;

/* HandlesMidi note events, block-granular (orchestrator.rs:856-859: handle_work
 * runs once per tick(), so events land at the start of a block). */
typedef struct {
  uint32_t voice;
  uint8_t key;
  uint8_t velocity;
  uint8_t on; /* 1 = note_on(key, vel); 0 = note_off(vel) */
  uint8_t reserved;
} groove_note_event;

/* Effects, settings/src/effects.rs:19-56. */
typedef enum {
  GROOVE_FX_GAIN = 0,        /* Gain{ceiling}                                   */
  GROOVE_FX_BITCRUSHER = 1,  /* Bitcrusher{bits}                                */
  GROOVE_FX_BIQUAD_LP12 = 2, /* filter-low-pass-12db{cutoff, q}                 */
  GROOVE_FX_BIQUAD_LP24 = 3, /* filter-low-pass-24db{cutoff, passband-ripple}   */
  GROOVE_FX_CHORUS = 4,      /* Chorus{voices, delay-seconds}                   */
  GROOVE_FX_DELAY = 5,       /* Delay{seconds}                                  */
  GROOVE_FX_REVERB = 6,      /* Reverb{attenuation, seconds}                    */
  GROOVE_FX_MIXER = 7,       /* Mixer (identity; orchestrator.rs:543-546)       */
  GROOVE_FX_BIQUAD_HP12 = 8, /* filter-high-pass-12db{cutoff, q}                */
  GROOVE_FX_LIMITER = 9,     /* Limiter{min, max}                               */
  GROOVE_FX_COMPRESSOR = 10, /* Compressor{threshold, ratio}                    */
  GROOVE_FX_BIQUAD_BP12 = 11,     /* filter-band-pass-12db{cutoff, bandwidth}    */
  GROOVE_FX_BIQUAD_BS12 = 12,     /* filter-band-stop-12db{cutoff, bandwidth}    */
  GROOVE_FX_BIQUAD_AP12 = 13,     /* filter-all-pass-12db{cutoff, q}             */
  GROOVE_FX_BIQUAD_PEAK12 = 14,   /* filter-peaking-eq-12db{cutoff, db-gain}     */
  GROOVE_FX_BIQUAD_LSHELF12 = 15, /* filter-low-shelf-12db{cutoff, db-gain}      */
  GROOVE_FX_BIQUAD_HSHELF12 = 16  /* filter-high-shelf-12db{cutoff, db-gain}     */
} groove_fx_kind;
#define GROOVE_FX_KIND_COUNT 17

/* Per-lane effect parameters.  One struct for every kind keeps the ABI flat;
 * unused fields are ignored.  Structural fields (marked UNIFORM) must be equal
 * across all lanes of one effect bank: they fix delay-line lengths, so the ring
 * index is wave-uniform and ring rows are coalesced. */
typedef struct {
  float ceiling;         /* Gain                                            */
  uint32_t bits;         /* Bitcrusher: bits to crush (0..15)                */
  float cutoff_hz;       /* BiQuad LP12 / LP24 / HP12                        */
  float q;               /* BiQuad LP12 / HP12                               */
  float passband_ripple; /* BiQuad LP24                                      */
  uint32_t voices;       /* Chorus: taps                (UNIFORM)            */
  float delay_seconds;   /* Chorus / Delay line length  (UNIFORM)            */
  float attenuation;     /* Reverb: input attenuation                        */
  float reverb_seconds;  /* Reverb: -60 dB decay time   (UNIFORM)            */
  float wet;             /* wet-dry-mix, 1.0 = fully wet (reference default) */
  float limit_min;       /* Limiter / Compressor threshold                   */
  float limit_max;       /* Limiter max / Compressor ratio                   */
  float bandwidth_hz;    /* BiQuad band-pass / band-stop: -3 dB bandwidth in Hz */
  float db_gain;         /* BiQuad peaking / shelves: gain in dB              */
} groove_fx_params;

/* Controllable indices for groove_bank_set_param / groove_fx_set_param
 * (#[derive(Control)] flattens params to an index→name table,
 * proc-macros/src/control.rs:171-249; names here are the kebab-case names). */
typedef enum {
  GROOVE_CTL_FX_CEILING = 0,         /* "ceiling"          */
  GROOVE_CTL_FX_BITS = 1,            /* "bits"             */
  GROOVE_CTL_FX_CUTOFF = 2,          /* "cutoff" (value01 → Hz via percent_to_frequency) */
  GROOVE_CTL_FX_Q = 3,               /* "q"                */
  GROOVE_CTL_FX_PASSBAND_RIPPLE = 4, /* "passband-ripple"  */
  GROOVE_CTL_FX_ATTENUATION = 5,     /* "attenuation"      */
  GROOVE_CTL_FX_WET = 6,             /* "wet-dry-mix"      */
  GROOVE_CTL_FX_THRESHOLD = 7,       /* "threshold" (Compressor; the reference's compressor demo ramps it) */
  GROOVE_CTL_WELSH_DCA_GAIN = 32,    /* "dca-gain"         */
  GROOVE_CTL_WELSH_DCA_PAN = 33,     /* "dca-pan"          */
  GROOVE_CTL_WELSH_CUTOFF = 34       /* "filter-cutoff" (value01 → Hz) */
} groove_control_index;

#ifdef __cplusplus
}
#endif
#endif /* GROOVE_TYPES_H */
