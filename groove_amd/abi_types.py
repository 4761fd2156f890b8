"""ctypes mirrors of include/groove_types.h (field order and types must match exactly).

Each structure corresponds to a `*Params` struct of the reference
(/root/reference/settings/src/patches.rs:110-164, 691-715; settings/src/effects.rs:19-56).
"""
import ctypes as C

DEFAULT_SAMPLE_RATE = 44100
BLOCK_FRAMES = 256
ALL_VOICES = 0xFFFFFFFF

# groove_waveform
WAVE_NONE, WAVE_SINE, WAVE_SQUARE, WAVE_PULSE_WIDTH, WAVE_TRIANGLE, WAVE_SAWTOOTH, WAVE_NOISE, \
    WAVE_DEBUG_ZERO, WAVE_DEBUG_MAX, WAVE_DEBUG_MIN, WAVE_TRIANGLE_SINE = range(11)
# groove_lfo_routing
LFO_NONE, LFO_AMPLITUDE, LFO_PITCH, LFO_PULSE_WIDTH, LFO_FILTER_CUTOFF, LFO_PITCH_OSC2, LFO_PW_OSC1, LFO_PW_OSC2, \
    LFO_RESONANCE, LFO_CUTOFF_AMP = range(10)
# groove_fx_kind
FX_GAIN, FX_BITCRUSHER, FX_BIQUAD_LP12, FX_BIQUAD_LP24, FX_CHORUS, FX_DELAY, FX_REVERB, FX_MIXER, \
    FX_BIQUAD_HP12, FX_LIMITER, FX_COMPRESSOR, FX_BIQUAD_BP12, FX_BIQUAD_BS12, FX_BIQUAD_AP12, FX_BIQUAD_PEAK12, \
    FX_BIQUAD_LSHELF12, FX_BIQUAD_HSHELF12 = range(17)
# groove_control_index
CTL_FX_CEILING, CTL_FX_BITS, CTL_FX_CUTOFF, CTL_FX_Q, CTL_FX_PASSBAND_RIPPLE, CTL_FX_ATTENUATION, CTL_FX_WET, CTL_FX_THRESHOLD = range(8)
CTL_WELSH_DCA_GAIN, CTL_WELSH_DCA_PAN, CTL_WELSH_CUTOFF = 32, 33, 34


class EnvelopeParams(C.Structure):
    _fields_ = [("attack", C.c_double), ("decay", C.c_double), ("sustain", C.c_double), ("release", C.c_double)]


class OscillatorParams(C.Structure):
    _fields_ = [("waveform", C.c_uint32), ("duty", C.c_float), ("tune", C.c_double), ("fixed_hz", C.c_double)]


class WelshParams(C.Structure):
    _fields_ = [
        ("oscillator_1", OscillatorParams), ("oscillator_2", OscillatorParams),
        ("oscillator_2_sync", C.c_uint32), ("oscillator_mix", C.c_float),
        ("amp_envelope", EnvelopeParams), ("filter_envelope", EnvelopeParams),
        ("lfo_waveform", C.c_uint32), ("lfo_routing", C.c_uint32), ("lfo_frequency", C.c_double),
        ("lfo_depth", C.c_float), ("filter_cutoff_hz", C.c_float), ("filter_passband_ripple", C.c_float),
        ("filter_cutoff_start", C.c_float), ("filter_cutoff_end", C.c_float),
        ("dca_gain", C.c_float), ("dca_pan", C.c_float),
    ]


class FmParams(C.Structure):
    _fields_ = [
        ("ratio", C.c_double), ("depth", C.c_float), ("beta", C.c_float),
        ("carrier_envelope", EnvelopeParams), ("modulator_envelope", EnvelopeParams),
        ("dca_gain", C.c_float), ("dca_pan", C.c_float),
    ]


class SampleDesc(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint32), ("root_hz", C.c_float)]


class SamplerParams(C.Structure):
    _fields_ = [("sample_index", C.c_uint32), ("one_shot", C.c_uint32), ("gain", C.c_float)]


class NoteEvent(C.Structure):
    _fields_ = [("voice", C.c_uint32), ("key", C.c_uint8), ("velocity", C.c_uint8), ("on", C.c_uint8),
                ("reserved", C.c_uint8)]


class FxParams(C.Structure):
    _fields_ = [
        ("ceiling", C.c_float), ("bits", C.c_uint32), ("cutoff_hz", C.c_float), ("q", C.c_float),
        ("passband_ripple", C.c_float), ("voices", C.c_uint32), ("delay_seconds", C.c_float),
        ("attenuation", C.c_float), ("reverb_seconds", C.c_float), ("wet", C.c_float),
        ("limit_min", C.c_float), ("limit_max", C.c_float), ("bandwidth_hz", C.c_float), ("db_gain", C.c_float),
    ]


def fx_params(**kw):
    """FxParams with the reference defaults (wet-dry-mix fully wet, gain ceiling 1)."""
    p = FxParams(ceiling=1.0, bits=8, cutoff_hz=1000.0, q=0.707, passband_ripple=0.707, voices=4,
                 delay_seconds=0.25, attenuation=0.95, reverb_seconds=1.25, wet=1.0, limit_min=0.0, limit_max=1.0,
                 bandwidth_hz=500.0, db_gain=6.0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def note_events(triples):
    """[(voice, key, on), ...] → ctypes array of NoteEvent (velocity 127)."""
    arr = (NoteEvent * len(triples))()
    for i, (voice, key, on) in enumerate(triples):
        arr[i].voice = voice
        arr[i].key = key
        arr[i].velocity = 127
        arr[i].on = 1 if on else 0
    return arr


def note_events_np(voices, keys, on):
    """Vectorised builder for large event lists (numpy arrays of equal length)."""
    import numpy as np
    n = len(voices)
    raw = np.zeros(n, dtype=np.dtype([("voice", "<u4"), ("key", "u1"), ("velocity", "u1"), ("on", "u1"), ("reserved", "u1")]))
    raw["voice"] = voices
    raw["key"] = keys
    raw["velocity"] = 127
    raw["on"] = 1 if on else 0
    arr = (NoteEvent * n).from_buffer_copy(raw.tobytes())
    return arr
