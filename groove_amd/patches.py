"""Synthetic, deterministic "Welsh-shaped" patches and projects.

The reference's 106 Welsh patch files are copyrighted data and are not copied; the table
below covers every waveform, LFO routing, sync / fixed-frequency / noise variant and the
parameter ranges of the shipped patches (cutoff 40 Hz-20 kHz, envelopes 0-30 s, the
`release := decay` quirk of /root/reference/settings/src/patches.rs:133-138).
No RNG: everything is a closed-form function of the voice index.
"""
import ctypes as C
import math

import numpy as np

from . import abi_types as T

N_PATCHES = 32


def note_to_frequency(key):
    return 440.0 * 2.0 ** ((key - 69) / 12.0)


def semis_and_cents(semis, cents):
    """OscillatorSettings::semis_and_cents, settings/src/patches.rs:255-258."""
    return 2.0 ** ((semis * 100.0 + cents) / 1200.0)


def percent_to_frequency(p):
    return 25.0 * 800.0 ** p


def frequency_to_percent(f):
    return math.log(f / 25.0) / math.log(800.0)


def denormalize_q(n):
    return n * n * 10.0 + 0.707


_PW = T.WAVE_PULSE_WIDTH
_O1 = [(_PW, .1), (T.WAVE_SQUARE, .5), (T.WAVE_SAWTOOTH, .5), (T.WAVE_TRIANGLE, .5),
       (_PW, .25), (T.WAVE_SQUARE, .5), (T.WAVE_SAWTOOTH, .5), (T.WAVE_TRIANGLE, .5),
       (_PW, .45), (T.WAVE_SQUARE, .5), (T.WAVE_SAWTOOTH, .5), (T.WAVE_SINE, .5),
       (_PW, .25), (T.WAVE_SQUARE, .5), (T.WAVE_SAWTOOTH, .5), (T.WAVE_TRIANGLE, .5),
       (_PW, .1), (T.WAVE_SQUARE, .5), (T.WAVE_NONE, .5), (T.WAVE_TRIANGLE, .5),
       (_PW, .45), (T.WAVE_SQUARE, .5), (T.WAVE_SAWTOOTH, .5), (T.WAVE_TRIANGLE_SINE, .5),
       (_PW, .25), (T.WAVE_SQUARE, .5), (T.WAVE_NONE, .5), (T.WAVE_SINE, .5),
       (_PW, .1), (_PW, .45), (_PW, .25), (_PW, .1)]
_O2 = [(T.WAVE_SAWTOOTH, .5), (T.WAVE_SQUARE, .5), (T.WAVE_TRIANGLE, .5), (_PW, .25),
       (T.WAVE_SINE, .5), (T.WAVE_NOISE, .5), (T.WAVE_NONE, .5), (T.WAVE_SAWTOOTH, .5)]
_ROUTING = [T.LFO_NONE, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE, T.LFO_PULSE_WIDTH, T.LFO_AMPLITUDE,
            T.LFO_PITCH, T.LFO_NONE, T.LFO_FILTER_CUTOFF, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE,
            T.LFO_NONE, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE, T.LFO_PULSE_WIDTH, T.LFO_AMPLITUDE,
            T.LFO_PITCH, T.LFO_NONE, T.LFO_NONE, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE,
            T.LFO_FILTER_CUTOFF, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE, T.LFO_PULSE_WIDTH,
            T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_NONE]
_LFO_WAVE = [T.WAVE_SINE, T.WAVE_TRIANGLE, T.WAVE_SINE, T.WAVE_SQUARE, T.WAVE_SINE, T.WAVE_SAWTOOTH, T.WAVE_SINE, T.WAVE_TRIANGLE]
_ATTACK = [0.0, 0.002, 0.01, 0.05, 0.2, 1.0, 0.06, 0.0]
_DECAY = [0.3, 0.05, 1.5, 3.29, 30.0, 0.6, 0.0, 0.12]
_SUSTAIN = [0.78, 1.0, 0.3, 0.6, 0.0, 0.9, 1.0, 0.45]


def welsh_patch(j):
    """Synthetic patch j (0..31) as a WelshParams (already in `derive_welsh_synth_params` form)."""
    j %= N_PATCHES
    p = T.WelshParams()
    w1, d1 = _O1[j]
    w2, d2 = _O2[(j * 3 + j // 8) % 8]
    if w1 == T.WAVE_NONE and w2 == T.WAVE_NONE:
        w2 = T.WAVE_SAWTOOTH
    p.oscillator_1.waveform, p.oscillator_1.duty, p.oscillator_1.tune, p.oscillator_1.fixed_hz = w1, d1, 1.0, 0.0
    tune_sel = j % 4
    tune2 = [1.0, semis_and_cents(12, 0.0), semis_and_cents(7, 5.0), 1.0][tune_sel]
    fixed2 = note_to_frequency(60) if (tune_sel == 3 and j % 8 == 7) else 0.0  # oscillator_2_track == false
    p.oscillator_2.waveform, p.oscillator_2.duty, p.oscillator_2.tune, p.oscillator_2.fixed_hz = w2, d2, tune2, fixed2
    p.oscillator_2_sync = 1 if j % 6 == 1 else 0
    # oscillator_mix, patches.rs:123-132
    if w1 == T.WAVE_NONE or w2 == T.WAVE_NONE:
        p.oscillator_mix = 1.0 if w2 == T.WAVE_NONE else 0.0
    else:
        m1, m2 = 1.0, [1.0, 0.5, 0.25, 0.75][j % 4]
        p.oscillator_mix = m1 / (m1 + m2)
    a, d, s = _ATTACK[j % 8], _DECAY[(j // 2) % 8], _SUSTAIN[(j // 3) % 8]
    if s == 0.0 and d < 0.3:
        d = 0.3
    p.amp_envelope = T.EnvelopeParams(a, d, s, d)  # release := decay (quirk)
    fa, fd, fs = _ATTACK[(j + 3) % 8], _DECAY[(j + 5) % 8], _SUSTAIN[(j + 1) % 8]
    p.filter_envelope = T.EnvelopeParams(fa, fd, fs, fd)
    p.lfo_waveform = _LFO_WAVE[j % 8]
    p.lfo_routing = _ROUTING[j]
    p.lfo_frequency = [0.53, 2.07, 5.13, 7.49][(j // 4) % 4]  # non-round: no exact phase ties at 44.1 kHz
    p.lfo_depth = [0.02, 0.05, 0.1, 0.2, 0.3, 0.5][j % 6]
    cutoff24 = 40.0 * 500.0 ** (((j * 11) % 32) / 31.0)
    cutoff12 = 40.0 * 500.0 ** (((j * 7 + 5) % 32) / 31.0)
    p.filter_cutoff_hz = cutoff24
    p.filter_passband_ripple = denormalize_q([0.0, 0.1, 0.3, 0.5][(j // 2) % 4])
    p.filter_cutoff_start = min(1.0, max(0.0, frequency_to_percent(cutoff12)))
    p.filter_cutoff_end = [0.0, 0.3, 0.6, 0.9][j % 4] if p.lfo_routing != T.LFO_FILTER_CUTOFF else 0.0
    p.dca_gain = 1.0
    p.dca_pan = ((j % 5) - 2) / 4.0
    return p



# ------------------------------------------------------------------ the reference library's class proportions (round 6)
# The workload `welsh-1m-library`: 106 synthetic patches whose CATEGORIES follow the reference's 106 shipped patch files, one slot per file:
# what the LFO is routed to, its waveform, how the filter is retuned (not at all / by its envelope / by the LFO), whether the filter's
# ripple is above ~3.8 (WF_COEF_WIDE), the two oscillators' waveforms, hard sync.  The COUNTS below are printed by
# tools/library_proportions.py (development container only: it reads the files where they lie; profiles/r06_library_proportions.json
# is its output); every VALUE — cutoffs, envelopes, LFO frequencies and depths, pans, tunings — is synthetic, from the formulas of
# welsh_patch().  The 32-patch benchmark table above pairs every edge-moving routing with a sine LFO and keeps every ripple below 3.3:
# it has no voice in the kernels a square / sawtooth / noise LFO on the pitch or a high-ripple swept filter asks for; 18 of the
# reference's 106 files do.
# (count, LFO routing, LFO waveform, filter retune: "static" | "env" | "lfo" | "reso", ripple above 3.8)
LIBRARY_LFO_ROWS = [
    (1, T.LFO_AMPLITUDE, T.WAVE_NOISE, "static", False), (1, T.LFO_AMPLITUDE, T.WAVE_SINE, "env", False),
    (2, T.LFO_AMPLITUDE, T.WAVE_SQUARE, "env", False), (1, T.LFO_AMPLITUDE, T.WAVE_SQUARE, "static", False),
    (15, T.LFO_AMPLITUDE, T.WAVE_TRIANGLE, "env", False), (1, T.LFO_AMPLITUDE, T.WAVE_TRIANGLE, "env", True),
    (6, T.LFO_AMPLITUDE, T.WAVE_TRIANGLE, "static", False),
    (1, T.LFO_FILTER_CUTOFF, T.WAVE_NOISE, "lfo", True), (1, T.LFO_FILTER_CUTOFF, T.WAVE_TRIANGLE, "lfo", True),
    (1, T.LFO_FILTER_CUTOFF, T.WAVE_TRIANGLE_SINE, "lfo", True), (1, T.LFO_CUTOFF_AMP, T.WAVE_TRIANGLE_SINE, "lfo", True),
    (21, T.LFO_NONE, T.WAVE_NONE, "env", False), (3, T.LFO_NONE, T.WAVE_NONE, "env", True), (12, T.LFO_NONE, T.WAVE_NONE, "static", False),
    (1, T.LFO_NONE, T.WAVE_SQUARE, "env", False),
    (1, T.LFO_PITCH, T.WAVE_NOISE, "env", False), (1, T.LFO_PITCH, T.WAVE_SAWTOOTH, "static", False),
    (3, T.LFO_PITCH, T.WAVE_SINE, "env", False), (1, T.LFO_PITCH, T.WAVE_SINE, "env", True), (1, T.LFO_PITCH, T.WAVE_SQUARE, "env", True),
    (4, T.LFO_PITCH, T.WAVE_TRIANGLE, "env", False), (1, T.LFO_PITCH, T.WAVE_TRIANGLE, "env", True),
    (9, T.LFO_PITCH, T.WAVE_TRIANGLE, "static", False), (1, T.LFO_PITCH, T.WAVE_TRIANGLE_SINE, "env", False),
    (2, T.LFO_PITCH_OSC2, T.WAVE_TRIANGLE, "static", False), (1, T.LFO_PITCH_OSC2, T.WAVE_TRIANGLE_SINE, "env", True),
    (2, T.LFO_PITCH_OSC2, T.WAVE_TRIANGLE_SINE, "static", False),
    (1, T.LFO_PULSE_WIDTH, T.WAVE_SAWTOOTH, "env", False), (1, T.LFO_PULSE_WIDTH, T.WAVE_SINE, "env", False),
    (1, T.LFO_PULSE_WIDTH, T.WAVE_TRIANGLE, "static", False), (2, T.LFO_PULSE_WIDTH, T.WAVE_TRIANGLE_SINE, "env", False),
    (2, T.LFO_PULSE_WIDTH, T.WAVE_TRIANGLE_SINE, "env", True), (1, T.LFO_PULSE_WIDTH, T.WAVE_TRIANGLE_SINE, "static", False),
    (1, T.LFO_PW_OSC1, T.WAVE_TRIANGLE, "env", False), (1, T.LFO_PW_OSC2, T.WAVE_TRIANGLE_SINE, "static", False),
    (1, T.LFO_RESONANCE, T.WAVE_SQUARE, "reso", False),
]
# (count, oscillator 1 waveform, oscillator 2 waveform)
LIBRARY_OSC_ROWS = [
    (21, _PW, _PW), (11, T.WAVE_NONE, T.WAVE_NONE), (10, T.WAVE_SQUARE, T.WAVE_SQUARE), (9, T.WAVE_TRIANGLE, T.WAVE_TRIANGLE),
    (6, T.WAVE_SQUARE, T.WAVE_SAWTOOTH), (6, T.WAVE_SAWTOOTH, T.WAVE_SQUARE), (5, _PW, T.WAVE_SQUARE), (5, T.WAVE_SQUARE, _PW),
    (4, T.WAVE_SAWTOOTH, _PW), (3, T.WAVE_TRIANGLE, T.WAVE_SQUARE), (3, T.WAVE_TRIANGLE, T.WAVE_NONE), (3, _PW, T.WAVE_NONE),
    (3, T.WAVE_SAWTOOTH, T.WAVE_SAWTOOTH), (3, _PW, T.WAVE_TRIANGLE), (3, T.WAVE_SAWTOOTH, T.WAVE_TRIANGLE), (2, T.WAVE_SAWTOOTH, T.WAVE_NONE),
    (2, T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH), (2, _PW, T.WAVE_SAWTOOTH), (1, T.WAVE_DEBUG_MAX, T.WAVE_NONE), (1, T.WAVE_NONE, T.WAVE_SQUARE),
    (1, T.WAVE_SQUARE, T.WAVE_TRIANGLE), (1, T.WAVE_SINE, T.WAVE_NONE), (1, T.WAVE_NONE, T.WAVE_SAWTOOTH),
]
LIBRARY_SLOTS = sum(r[0] for r in LIBRARY_LFO_ROWS)   # 106
assert LIBRARY_SLOTS == 106 and sum(r[0] for r in LIBRARY_OSC_ROWS) == LIBRARY_SLOTS
LIBRARY_SYNC_SLOTS = 18      # hard sync: 18 of the 106 files
_LIB_LFO = [r[1:] for r in LIBRARY_LFO_ROWS for _ in range(r[0])]
_LIB_OSC = [r[1:] for r in LIBRARY_OSC_ROWS for _ in range(r[0])]


def library_patch(s):
    """Synthetic patch of library slot s (0..105): the categories of LIBRARY_*_ROWS, the values of welsh_patch()'s formulas."""
    s %= LIBRARY_SLOTS
    routing, lfo_wave, retune, wide = _LIB_LFO[s]
    w1, w2 = _LIB_OSC[(s * 37 + 11) % LIBRARY_SLOTS]   # (37 is coprime with 106: the pairs are spread over the LFO rows)
    # (the reference's 11 none x none files are silent in its own derivation too — patches.rs:88-109 pushes no oscillator; here the
    # none x none slots sound a noise source, what 14 of the files mix in, so that every slot of the benchmark does its work)
    if w1 == T.WAVE_NONE and w2 == T.WAVE_NONE:
        w2 = T.WAVE_NOISE
    p = T.WelshParams()
    d1 = [.1, .25, .45, .3][s % 4] if w1 == _PW else .5
    d2 = [.25, .45, .1, .35][(s // 2) % 4] if w2 == _PW else .5
    p.oscillator_1.waveform, p.oscillator_1.duty, p.oscillator_1.tune, p.oscillator_1.fixed_hz = w1, d1, 1.0, 0.0
    tune2 = [1.0, semis_and_cents(12, 0.0), semis_and_cents(7, 5.0), semis_and_cents(-12, 4.0), 1.0, semis_and_cents(0, 7.0)][s % 6]   # (an octave below, 4 cents sharp: exactly an octave below key 57 is 110 Hz, whose edges tie every 2,205 frames — docs/DSP_SPEC.md section 2)
    fixed2 = note_to_frequency(60) if (s % 35 == 17 and w2 != T.WAVE_NONE) else 0.0   # oscillator_2_track == false: 3 of the files
    p.oscillator_2.waveform, p.oscillator_2.duty, p.oscillator_2.tune, p.oscillator_2.fixed_hz = w2, d2, tune2, fixed2
    p.oscillator_2_sync = 1 if (s % 6 == 1 and w1 != T.WAVE_NONE and w2 != T.WAVE_NONE) else 0   # (18 slots have s mod 6 == 1)
    if w1 == T.WAVE_NONE or w2 == T.WAVE_NONE:
        p.oscillator_mix = 1.0 if w2 == T.WAVE_NONE else 0.0
    else:
        p.oscillator_mix = 1.0 / (1.0 + [1.0, 0.5, 0.25, 0.75][s % 4])
    a, d, su = _ATTACK[s % 8], _DECAY[(s // 2) % 8], _SUSTAIN[(s // 3) % 8]
    if su == 0.0 and d < 0.3:
        d = 0.3
    p.amp_envelope = T.EnvelopeParams(a, d, su, d)
    fa, fd, fs = _ATTACK[(s + 3) % 8], _DECAY[(s + 5) % 8], _SUSTAIN[(s + 1) % 8]
    p.filter_envelope = T.EnvelopeParams(fa, fd, fs, fd)
    p.lfo_waveform, p.lfo_routing = lfo_wave, routing
    p.lfo_frequency = [0.53, 2.07, 5.13, 7.49, 2.41, 4.03, 0.71, 7.53, 31.7, 1.03][(s // 3) % 10]   # non-round: no exact phase ties at 44.1 kHz
    p.lfo_depth = [0.02, 0.05, 0.1, 0.2, 0.3, 0.5][s % 6]
    if routing == T.LFO_RESONANCE:
        p.lfo_depth = 0.5
    cutoff24 = 40.0 * 500.0 ** (((s * 11) % 32) / 31.0)
    cutoff12 = 40.0 * 500.0 ** (((s * 7 + 5) % 32) / 31.0)
    p.filter_cutoff_hz = cutoff24
    # ripple = denormalize_q(filter-resonance): 61 of the files at 0 (0.707), the rest up to 1 (10.7); above ~0.556 (3.8) it is "wide"
    p.filter_passband_ripple = denormalize_q([0.6, 0.65, 0.7, 0.8, 1.0][s % 5] if wide else [0.0, 0.0, 0.0, 0.1, 0.3, 0.0, 0.45, 0.5][(s // 2) % 8])
    p.filter_cutoff_start = min(1.0, max(0.0, frequency_to_percent(cutoff12)))
    p.filter_cutoff_end = [0.3, 0.6, 0.9, 0.5][s % 4] if retune == "env" else 0.0
    p.dca_gain = 1.0
    p.dca_pan = ((s % 5) - 2) / 4.0
    return p


def library_table():
    return (T.WelshParams * LIBRARY_SLOTS)(*[library_patch(s) for s in range(LIBRARY_SLOTS)])


# The Welsh patch tables of the synthetic workloads (groove_amd/projects.py): name -> (entries, patch function).  Voice w plays entry w mod entries.
PATCH_TABLES = {"benchmark-32": (N_PATCHES, welsh_patch), "library-106": (LIBRARY_SLOTS, library_patch)}


def random_welsh_patch(rng):
    """A Welsh patch with every continuous parameter DRAWN (numpy Generator `rng`) instead of taken from the 32-entry benchmark table: any
    waveform pair, duty 0.05 - 0.95, oscillator 2 an octave either way (or at a fixed pitch), hard sync, envelopes with instant attacks and
    zero sustains among them, every LFO routing and waveform, cutoffs 40 Hz - 20 kHz at ripples 0.71 - 4.3, sweeps of any extent.  For the
    randomised parity tests (tests/test_gpu_random_inputs.py) and tools/random_patch_probe.py."""
    waves = [T.WAVE_NONE, T.WAVE_SINE, T.WAVE_SQUARE, T.WAVE_PULSE_WIDTH, T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH, T.WAVE_NOISE]
    p = T.WelshParams()
    w1, w2 = int(rng.choice(waves)), int(rng.choice(waves))
    if w1 == T.WAVE_NONE and w2 == T.WAVE_NONE:
        w2 = T.WAVE_SAWTOOTH
    p.oscillator_1.waveform, p.oscillator_1.duty, p.oscillator_1.tune, p.oscillator_1.fixed_hz = w1, float(rng.uniform(0.05, 0.95)), 1.0, 0.0
    p.oscillator_2.waveform, p.oscillator_2.duty = w2, float(rng.uniform(0.05, 0.95))
    p.oscillator_2.tune = float(2.0 ** rng.uniform(-1.0, 1.0))
    p.oscillator_2.fixed_hz = float(rng.uniform(100.0, 2000.0)) if rng.random() < 0.15 else 0.0
    p.oscillator_2_sync = int(rng.random() < 0.25)
    p.oscillator_mix = 1.0 if w2 == T.WAVE_NONE else 0.0 if w1 == T.WAVE_NONE else float(rng.uniform(0.1, 0.9))

    def env():
        return T.EnvelopeParams(0.0 if rng.random() < 0.2 else float(rng.uniform(0.001, 0.3)), float(rng.uniform(0.05, 2.0)),
                                0.0 if rng.random() < 0.15 else float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.05, 1.5)))
    p.amp_envelope, p.filter_envelope = env(), env()
    if p.amp_envelope.sustain == 0.0 and p.amp_envelope.decay < 0.3:
        p.amp_envelope.decay = 0.3
    p.lfo_waveform = int(rng.choice([T.WAVE_SINE, T.WAVE_SQUARE, T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH, T.WAVE_PULSE_WIDTH]))
    p.lfo_routing = int(rng.integers(0, 10))
    p.lfo_frequency = float(rng.uniform(0.1, 12.0))
    p.lfo_depth = float(rng.uniform(0.0, 0.5))
    p.filter_cutoff_hz = float(40.0 * 500.0 ** rng.uniform(0.0, 1.0))
    p.filter_passband_ripple = denormalize_q(float(rng.uniform(0.0, 0.6)))
    p.filter_cutoff_start = float(rng.uniform(0.0, 1.0))
    p.filter_cutoff_end = 0.0 if p.lfo_routing == T.LFO_FILTER_CUTOFF or rng.random() < 0.3 else float(rng.uniform(0.0, 1.0))
    p.dca_gain, p.dca_pan = float(rng.uniform(0.3, 1.0)), float(rng.uniform(-1.0, 1.0))
    return p

def _tile(table, n, ctype):
    """Tile a short ctypes table to n entries by index modulo (vectorised via numpy)."""
    size = C.sizeof(ctype)
    raw = np.frombuffer(bytes(bytearray(table)), dtype=np.uint8).reshape(len(table), size)
    idx = np.arange(n, dtype=np.int64) % len(table)
    out = np.ascontiguousarray(raw[idx])
    return (ctype * n).from_buffer_copy(out.tobytes())


def welsh_voices(n, first_voice=0):
    """Config #2 rule: voice i uses patch (i mod 32)."""
    table = (T.WelshParams * N_PATCHES)(*[welsh_patch((j + first_voice) % N_PATCHES) for j in range(N_PATCHES)])
    return _tile(table, n, T.WelshParams)


def grouped_order(n, first_voice=0):
    """Project voice indices re-ordered synth-major: all voices of patch 0, then patch 1, ...
    (stable).  The reference's WelshSynth is ONE patch plus a voice store, so a many-voice
    project is a set of synths; laying the bank out synth by synth keeps every 64-lane
    wavefront on one patch.  Returns the original voice index of each lane."""
    i = np.arange(first_voice, first_voice + n, dtype=np.int64)
    order = np.argsort(i % N_PATCHES, kind="stable")
    return i[order]


def welsh_voices_grouped(n, first_voice=0):
    """Same multiset of voices as welsh_voices(n, first_voice), laid out synth-major."""
    idx = grouped_order(n, first_voice)
    table = (T.WelshParams * N_PATCHES)(*[welsh_patch(j) for j in range(N_PATCHES)])
    size = C.sizeof(T.WelshParams)
    raw = np.frombuffer(bytes(bytearray(table)), dtype=np.uint8).reshape(N_PATCHES, size)
    out = np.ascontiguousarray(raw[idx % N_PATCHES])
    return (T.WelshParams * n).from_buffer_copy(out.tobytes()), idx


def grouped_note_events(idx, on=True):
    """Note events for a grouped bank: lane L plays the key of original voice idx[L]."""
    keys = (36 + (7 * idx) % 49).astype(np.uint8)
    return T.note_events_np(np.arange(len(idx), dtype=np.uint32), keys, on)


def voice_keys(n, first_voice=0):
    """key = 36 + (7 i mod 49)."""
    i = np.arange(first_voice, first_voice + n, dtype=np.int64)
    return (36 + (7 * i) % 49).astype(np.uint8)


def note_on_all(n, first_voice=0):
    return T.note_events_np(np.arange(n, dtype=np.uint32), voice_keys(n, first_voice), True)


def note_off_all(n, first_voice=0):
    return T.note_events_np(np.arange(n, dtype=np.uint32), voice_keys(n, first_voice), False)


NOTE_OFF_FRAME = 22016  # config #2: note-off at frame 22,016 (= block 86)
RENDER_BLOCKS = 172     # 172 blocks x 256 = 44,032 frames


def fm_patch(j):
    """Config #5 FM voices: ratio 2, depth 1, beta in {0.1, 1, 10, 15}."""
    p = T.FmParams()
    p.ratio = 2.0
    p.depth = 1.0
    p.beta = [0.1, 1.0, 10.0, 15.0][j % 4]
    p.carrier_envelope = T.EnvelopeParams(_ATTACK[j % 8], _DECAY[(j // 2) % 8] or 0.3, [0.78, 1.0, 0.3, 0.6][j % 4], 0.3)
    p.modulator_envelope = T.EnvelopeParams(_ATTACK[(j + 2) % 8], _DECAY[(j + 1) % 8] or 0.3, [1.0, 0.5, 0.8, 0.2][j % 4], 0.5)
    p.dca_gain = 1.0
    p.dca_pan = ((j % 5) - 2) / 4.0
    return p


def fm_voices(n, first_voice=0):
    table = (T.FmParams * 16)(*[fm_patch((j + first_voice) % 16) for j in range(16)])
    return _tile(table, n, T.FmParams)


# Config #4: synthetic 60-buffer mono drum bank with the real 707 bank's size class
# (27,132 ... 87,705 frames, total 2,779,555 in the reference; lengths here are a fixed
# arithmetic progression with the same min / max / count).
BANK_BUFFERS = 60


def drum_bank(sample_rate=T.DEFAULT_SAMPLE_RATE, scale=1.0):
    """Returns (pcm float32 [total], descs (SampleDesc*60), lengths).  `scale` < 1 shrinks
    the buffers for fast CPU tests."""
    lengths = [int((27132 + (87705 - 27132) * k / (BANK_BUFFERS - 1)) * scale) for k in range(BANK_BUFFERS)]
    total = sum(lengths)
    pcm = np.empty(total, dtype=np.float32)
    descs = (T.SampleDesc * BANK_BUFFERS)()
    off = 0
    lcg = np.uint32(12345)
    for k, ln in enumerate(lengths):
        n = np.arange(ln, dtype=np.float64)
        f_k = 55.0 * 2.0 ** (k / 12.0)
        tau = 0.05 * sample_rate * (1 + (k % 7))
        tone = np.sin(2 * np.pi * f_k * n / sample_rate) * np.exp(-n / tau)
        # LCG noise, deterministic
        seq = (np.arange(1, ln + 1, dtype=np.uint64) * np.uint64(1664525) + np.uint64(1013904223 + k)) & np.uint64(0xFFFFFFFF)
        seq = (seq * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        noise = (seq.astype(np.float64) / 2147483648.0 - 1.0) * np.exp(-n / (0.3 * tau)) * 0.25
        pcm[off:off + ln] = (0.8 * tone + noise).astype(np.float32)
        descs[k].offset, descs[k].length = off, ln
        descs[k].root_hz = 0.0 if k % 2 == 0 else 440.0
        off += ln
    return pcm, descs, lengths


def sampler_voices(n):
    """Config #4: voice i plays buffer i mod 60; even i = drumkit one-shot (step 1), odd i =
    pitched sampler buffer (root 440 Hz; key chosen so step = 2^((i mod 25 - 12)/12))."""
    arr = (T.SamplerParams * n)()
    raw = np.zeros(n, dtype=np.dtype([("sample_index", "<u4"), ("one_shot", "<u4"), ("gain", "<f4")]))
    i = np.arange(n)
    raw["sample_index"] = i % BANK_BUFFERS
    raw["one_shot"] = 1
    raw["gain"] = 1.0
    C.memmove(arr, raw.tobytes(), n * C.sizeof(T.SamplerParams))
    return arr


def sampler_keys(n):
    """MIDI key per voice: 69 + (i mod 25 - 12) so that a 440 Hz-rooted buffer steps by 2^((i mod 25 - 12)/12)."""
    i = np.arange(n, dtype=np.int64)
    return (69 + (i % 25) - 12).astype(np.uint8)


def sampler_start_block(n):
    """start frame = (h(i) mod 172) * 256, h(i) = (i * 2654435761) mod 2^32."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return (h % np.uint64(172)).astype(np.int64)


def chain_fx_params(n, voice_index=None):
    """Config #3 per-voice chain: BiQuad LP12 (cutoff 1000 + 50 (i mod 64) Hz, q 0.707) → Chorus
    (4 voices, 0.25 s) → Delay (0.1 s) → Reverb (0.95, 1.25 s).  voice_index[lane] is the project
    voice index i of each lane (identity when None)."""
    def arr(**kw):
        a = (T.FxParams * n)()
        base = T.fx_params(**kw)
        raw = np.frombuffer(bytes(bytearray(base)), dtype=np.uint8)
        C.memmove(a, np.tile(raw, n).tobytes(), n * C.sizeof(T.FxParams))
        return a
    lp = arr(q=0.707)
    for i in range(n):
        lp[i].cutoff_hz = 1000.0 + 50.0 * (int(i if voice_index is None else voice_index[i]) % 64)
    return [(T.FX_BIQUAD_LP12, lp), (T.FX_CHORUS, arr(voices=4, delay_seconds=0.25)),
            (T.FX_DELAY, arr(delay_seconds=0.1)), (T.FX_REVERB, arr(attenuation=0.95, reverb_seconds=1.25))]
