"""CPU tier: the oracle against every known-answer the reference's own tests and in-tree
formula documents hold for this path.  No GPU."""
import math

import numpy as np
import pytest

from groove_amd import abi_types as T


# ---- settings/src/patches.rs:754-796 oscillator_tuning_helpers --------------------------
def test_semis_and_cents_and_octaves(oracle):
    L = oracle.lib()
    assert L.oracle_octaves(0) == 1.0 and L.oracle_octaves(1) == 2.0 and L.oracle_octaves(-1) == 0.5
    assert L.oracle_octaves(2) == 4.0 and L.oracle_octaves(-2) == 0.25
    assert L.oracle_semis_and_cents(0, 0.0) == 1.0
    assert L.oracle_semis_and_cents(12, 0.0) == 2.0
    assert abs(L.oracle_semis_and_cents(5, 0.0) - 1.3348398541700344) < 1e-15  # F4 / C4
    assert L.oracle_semis_and_cents(0, -100.0) == 2.0 ** (-100.0 / 1200.0)
    assert L.oracle_octaves(1) == L.oracle_semis_and_cents(12, 0.0) == L.oracle_semis_and_cents(0, 1200.0)
    assert L.oracle_semis_and_cents(1, 0.0) == L.oracle_semis_and_cents(0, 100.0)
    # note_to_frequency: A4 = 440, C4 = 261.6256 (patches.rs:778 comment)
    assert L.oracle_note_to_frequency(69) == 440.0
    assert abs(L.oracle_note_to_frequency(60) - 261.6255653) < 1e-6
    assert abs(L.oracle_note_to_frequency(65) / L.oracle_note_to_frequency(60) - 1.3348398541700344) < 1e-12


# ---- orchestration/src/orchestrator.rs:1444-1473 gather_audio_basic ----------------------
def test_gather_audio_basic(oracle):
    g = oracle.Graph()
    l1, l2 = g.add_source(0.1), g.add_source(0.2)
    assert not g.gather(1).any()                                  # nothing connected: silence
    assert g.patch(l1, g.MAIN_MIXER) == 0
    assert np.allclose(g.gather(1), 0.1, atol=1e-12)
    g.unpatch_all(); g.patch(l2, g.MAIN_MIXER)
    assert np.allclose(g.gather(1), 0.2, atol=1e-12)
    g.unpatch_all(); g.patch(l1, g.MAIN_MIXER); g.patch(l2, g.MAIN_MIXER)
    assert np.allclose(g.gather(1), 0.1 + 0.2, atol=1e-12)


# ---- orchestrator.rs:1475-1542 gather_audio (Gain, siblings, order independence) ----------
def test_gather_audio_gain_and_siblings(oracle):
    g = oracle.Graph()
    l1 = g.add_source(0.1)
    gain = g.add_effect(T.FX_GAIN, T.fx_params(ceiling=0.5))
    l2, l3, l4 = g.add_source(0.2), g.add_source(0.3), g.add_source(0.4)
    assert g.patch_chain_to_main_mixer([l1, gain]) == 0
    assert np.allclose(g.gather(1), 0.1 * 0.5, atol=1e-12)
    for l in (l2, l3, l4):
        g.patch(l, g.MAIN_MIXER)
    assert np.allclose(g.gather(1), 0.1 * 0.5 + 0.2 + 0.3 + 0.4, atol=1e-12)
    # same graph patched in the opposite order
    g2 = oracle.Graph()
    l4b, l3b, l2b = g2.add_source(0.4), g2.add_source(0.3), g2.add_source(0.2)
    gainb = g2.add_effect(T.FX_GAIN, T.fx_params(ceiling=0.5)); l1b = g2.add_source(0.1)
    for l in (l4b, l3b, l2b):
        g2.patch(l, g2.MAIN_MIXER)
    g2.patch_chain_to_main_mixer([l1b, gainb])
    assert np.allclose(g2.gather(1), g.gather(1), atol=1e-12)
    # a lone effect with no input contributes silence
    g3 = oracle.Graph()
    e = g3.add_effect(T.FX_GAIN, T.fx_params(ceiling=0.5)); g3.patch(e, g3.MAIN_MIXER)
    assert not g3.gather(1).any()
    # instruments have no inputs (test-data/instruments-have-no-inputs.json5)
    assert g3.patch(e, g3.add_source(0.3)) != 0


# ---- orchestrator.rs:1544-1640 gather_audio_2 (chains) -------------------------------------
def test_gather_audio_chains(oracle):
    g = oracle.Graph()
    def chain(level, *ceilings):
        uids = [g.add_source(level)] + [g.add_effect(T.FX_GAIN, T.fx_params(ceiling=c)) for c in ceilings]
        return uids
    c1, c2, c3 = chain(0.1, 0.2, 0.4), chain(0.3, 0.6), chain(0.5, 0.8)
    g.patch_chain_to_main_mixer(c1)
    f = lambda v: float(np.float32(v))  # ceilings are f32 fields of the params struct
    assert g.gather(1)[0, 0] == 0.1 * f(0.2) * f(0.4)  # exact (assert_eq at orchestrator.rs:1608)
    g.patch_chain_to_main_mixer(c2); g.patch_chain_to_main_mixer(c3)
    want = 0.1 * f(0.2) * f(0.4) + 0.3 * f(0.6) + 0.5 * f(0.8)
    assert np.allclose(g.gather(1), want, atol=1e-12)


# ---- orchestrator.rs:1642-1668 gather_audio_with_branches ----------------------------------
def test_gather_audio_with_branches(oracle):
    g = oracle.Graph()
    a = g.add_source(0.1)
    b, c = g.add_source(0.3), g.add_source(0.5)
    gain = g.add_effect(T.FX_GAIN, T.fx_params(ceiling=0.5))
    g.patch(a, g.MAIN_MIXER)
    g.patch(b, gain); g.patch(c, gain); g.patch(gain, g.MAIN_MIXER)
    assert np.allclose(g.gather(1), 0.1 + 0.5 * (0.3 + 0.5), atol=1e-12)
    # ToyEffect negates (util.rs tests)
    neg = g.add_toy_effect(); src = g.add_source(0.25)
    g.patch(src, neg); g.patch(neg, g.MAIN_MIXER)
    assert np.allclose(g.gather(1), 0.1 + 0.5 * (0.3 + 0.5) - 0.25, atol=1e-12)


# ---- orchestrator.rs:1689-1737, 1822-1827, 1903-1908 render lengths -------------------------
def test_render_frame_counts(oracle):
    L = oracle.lib()
    assert L.oracle_run_frames(0.0, 128.0, 44100.0, 64) == 0                       # zero timer
    assert L.oracle_run_frames(4.0, 240.0, 24000.0, 64) == 24000                   # ordinary timer
    assert L.oracle_run_frames(4.0, 128.0, 44100.0, 64) == math.ceil(4 * 60 / 128 * 44100) == 82688
    # run_performance drops the final partial block (orchestrator.rs:827-836), run keeps it (:795)
    assert L.oracle_run_performance_frames(4.0, 128.0, 44100.0, 64) == 82688 - 82688 % 64
    assert L.oracle_run_frames(8.0, 128.0, 44100.0, 256) == 165375                 # config #1: 2 measures


# ---- orchestration/src/util.rs:286-318 MMA transforms ---------------------------------------
def test_mma_transforms(oracle):
    L = oracle.lib()
    assert L.oracle_mma_concave(0.0) == 0.0 and L.oracle_mma_concave(1.0) == 1.0
    assert L.oracle_mma_convex(0.0) == 0.0 and L.oracle_mma_convex(1.0) == 1.0
    for x in np.linspace(0.01, 0.99, 50):
        assert L.oracle_mma_concave(x) <= x + 1e-12 or x > 0.99   # concave lies under the diagonal until the clip
        assert L.oracle_mma_convex(x) >= x - 1e-12
    assert L.oracle_mma_concave(0.999999) == 1.0
    assert L.oracle_mma_convex(1e-6) == 0.0
    assert abs(L.oracle_mma_concave(0.5) - (-(5 / 12) * math.log10(0.5))) < 1e-15


# ---- doc/Audio-EQ-Cookbook.txt:76-111, Eq 4 ---------------------------------------------------
def test_rbj_cookbook_lowpass(oracle):
    L = oracle.lib()
    import ctypes as C
    out = (C.c_double * 5)()
    for f0, q in ((1000.0, 0.707), (40.0, 0.707), (12000.0, 2.0), (4200.0, 10.0)):
        L.oracle_rbj_lowpass(f0, q, 44100.0, out)
        w0 = 2 * math.pi * f0 / 44100.0
        alpha = math.sin(w0) / (2 * q); a0 = 1 + alpha
        want = [(1 - math.cos(w0)) / 2 / a0, (1 - math.cos(w0)) / a0, (1 - math.cos(w0)) / 2 / a0, -2 * math.cos(w0) / a0, (1 - alpha) / a0]
        assert np.allclose(list(out), want, rtol=1e-14)
        b0, b1, b2, a1, a2 = out
        assert abs((b0 + b1 + b2) / (1 + a1 + a2) - 1.0) < 1e-9               # DC gain 1 (H(s) = 1/(s^2+s/Q+1) at s = 0)
        z = np.exp(1j * w0)
        H = (b0 + b1 / z + b2 / z ** 2) / (1 + a1 / z + a2 / z ** 2)
        assert abs(abs(H) - q) < 1e-9                                        # |H(j w0)| = Q
    # Direct Form 1 (Eq 4) impulse response against a literal evaluation of the recurrence
    L.oracle_rbj_lowpass(1000.0, 0.707, 44100.0, out)
    x = np.zeros(64); x[0] = 1.0
    y = np.zeros(64)
    dp = C.POINTER(C.c_double)
    L.oracle_biquad_df1_run(out, x.ctypes.data_as(dp), y.ctypes.data_as(dp), 64)
    b0, b1, b2, a1, a2 = out
    ref = np.zeros(64)
    for n in range(64):
        ref[n] = b0 * x[n] + (b1 * x[n - 1] if n >= 1 else 0) + (b2 * x[n - 2] if n >= 2 else 0) \
                 - (a1 * ref[n - 1] if n >= 1 else 0) - (a2 * ref[n - 2] if n >= 2 else 0)
    assert np.allclose(y, ref, atol=1e-15)
    L.oracle_rbj_highpass(1000.0, 0.707, 44100.0, out)
    b0, b1, b2, a1, a2 = out
    assert abs(b0 + b1 + b2) < 1e-12                                         # HPF blocks DC
    assert abs((b0 - b1 + b2) / (1 - a1 + a2) - 1.0) < 1e-9                  # unity at Nyquist


# ---- orchestration/src/helpers.rs:79-91 WAV quantisation ---------------------------------------
def test_wav_quantise(oracle):
    L = oracle.lib()
    assert [L.oracle_wav_quantise(v) for v in (0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0)] == [0, 32767, -32767, 16383, -16383, 32767, -32768]
    assert L.oracle_wav_quantise(float("nan")) == 0
    assert L.oracle_wav_quantise(0.99999) == int(0.99999 * 32767.0)
    assert L.oracle_wav_quantise(-3.0e-5) == 0                                # truncation toward zero, not floor


# ---- settings/src/patches.rs:925-936 welsh_makes_any_sound_at_all -------------------------------
def test_welsh_makes_any_sound_at_all(oracle):
    p = (T.WelshParams * 1)()
    p[0].oscillator_1.waveform = T.WAVE_SAWTOOTH; p[0].oscillator_1.tune = 1.0
    p[0].oscillator_2.waveform = T.WAVE_NONE; p[0].oscillator_2.tune = 1.0
    p[0].oscillator_mix = 1.0
    p[0].amp_envelope = T.EnvelopeParams(0.06, 30.0, 1.0, 0.3)
    p[0].filter_envelope = T.EnvelopeParams(0.0, 3.29, 0.78, 30.0)
    p[0].filter_cutoff_hz = 900.0; p[0].filter_passband_ripple = 0.707
    p[0].filter_cutoff_start = 0.1; p[0].filter_cutoff_end = 0.9
    p[0].dca_gain = 1.0
    b = oracle.Bank.welsh(p)
    assert not b.render(4).any()
    b.note_events(T.note_events([(0, 60, True)]))
    out = b.render(6)
    assert out[:, 5, 0].any(), "once triggered, the voice makes a sound within 5 frames"


# ---- MusicalTime / transport frame arithmetic: 1 s of frames at 60 bpm = 1 beat ---------------
def test_frames_per_beat(oracle):
    L = oracle.lib()
    for sr in (2000, 8000, 22050, 24000, 44100, 48000, 88200, 96000, 192000):
        assert L.oracle_performance_total_frames(1.0, 60.0, float(sr)) == sr


# ---- doc/Audio-EQ-Cookbook.txt:113-198: the remaining biquad modes ------------------------------
def _H(c, f, fs=44100.0):
    b0, b1, b2, a1, a2 = c
    z = np.exp(1j * 2 * np.pi * f / fs)
    return (b0 + b1 / z + b2 / z ** 2) / (1 + a1 / z + a2 / z ** 2)


def test_rbj_cookbook_other_modes(oracle):
    import ctypes as C
    L = oracle.lib()
    out = (C.c_double * 5)()
    fs, f0 = 44100.0, 1000.0
    w0 = 2 * math.pi * f0 / fs
    cw, sw = math.cos(w0), math.sin(w0)

    def coef(kind, **kw):
        p = T.fx_params(cutoff_hz=f0, **kw)
        assert L.oracle_rbj_for_kind(kind, C.byref(p), fs, out) == 1
        return list(out)

    # band-pass (constant 0 dB peak gain), bandwidth given in Hz → octaves between the -3 dB points
    bw_oct = math.log2((f0 + 150) / (f0 - 150))
    alpha = sw * math.sinh(math.log(2) / 2 * bw_oct * w0 / sw)
    c = coef(T.FX_BIQUAD_BP12, bandwidth_hz=300.0)
    assert np.allclose(c, [alpha / (1 + alpha), 0.0, -alpha / (1 + alpha), -2 * cw / (1 + alpha), (1 - alpha) / (1 + alpha)], rtol=1e-12, atol=1e-15)
    assert abs(abs(_H(c, f0)) - 1.0) < 1e-9 and abs(_H(c, 0.0)) < 1e-9
    # band-stop (notch): zero at f0, unity at DC and Nyquist
    c = coef(T.FX_BIQUAD_BS12, bandwidth_hz=300.0)
    assert np.allclose(c, [1 / (1 + alpha), -2 * cw / (1 + alpha), 1 / (1 + alpha), -2 * cw / (1 + alpha), (1 - alpha) / (1 + alpha)], rtol=1e-12)
    assert abs(_H(c, f0)) < 1e-9 and abs(abs(_H(c, 0.0)) - 1) < 1e-9 and abs(abs(_H(c, fs / 2)) - 1) < 1e-9
    # all-pass: |H| = 1 everywhere
    c = coef(T.FX_BIQUAD_AP12, q=0.707)
    a = sw / (2 * 0.707)
    a = sw / (2 * float(np.float32(0.707)))
    assert np.allclose(c, [(1 - a) / (1 + a), -2 * cw / (1 + a), 1.0, -2 * cw / (1 + a), (1 - a) / (1 + a)], rtol=1e-12)
    for f in (50.0, 1000.0, 9000.0):
        assert abs(abs(_H(c, f)) - 1.0) < 1e-9
    # peaking EQ: gain dBgain at f0, unity far away
    c = coef(T.FX_BIQUAD_PEAK12, db_gain=6.0)
    A = 10 ** (6.0 / 40)
    a = sw / (2 * math.sqrt(0.5))
    assert np.allclose(c, [(1 + a * A) / (1 + a / A), -2 * cw / (1 + a / A), (1 - a * A) / (1 + a / A), -2 * cw / (1 + a / A), (1 - a / A) / (1 + a / A)], rtol=1e-12)
    assert abs(20 * math.log10(abs(_H(c, f0))) - 6.0) < 1e-9 and abs(abs(_H(c, 0.0)) - 1) < 1e-9
    # shelves (S = 1): low shelf has gain A^2 at DC and 1 at Nyquist; high shelf the reverse
    c = coef(T.FX_BIQUAD_LSHELF12, db_gain=6.0)
    assert abs(20 * math.log10(abs(_H(c, 0.0))) - 6.0) < 1e-9 and abs(abs(_H(c, fs / 2)) - 1) < 1e-9
    t = 2 * math.sqrt(A) * (sw / 2 * math.sqrt(2))
    a0 = (A + 1) + (A - 1) * cw + t
    assert abs(c[0] - A * ((A + 1) - (A - 1) * cw + t) / a0) < 1e-14 and abs(c[3] - (-2 * ((A - 1) + (A + 1) * cw)) / a0) < 1e-14
    c = coef(T.FX_BIQUAD_HSHELF12, db_gain=6.0)
    assert abs(20 * math.log10(abs(_H(c, fs / 2))) - 6.0) < 1e-9 and abs(abs(_H(c, 0.0)) - 1) < 1e-9
    # a bandwidth wider than 2 f0 is capped (8 octaves) instead of producing NaN
    c = coef(T.FX_BIQUAD_BP12, bandwidth_hz=2000.0)
    assert np.isfinite(c).all()
    p = T.fx_params()
    assert L.oracle_rbj_for_kind(T.FX_GAIN, C.byref(p), fs, out) == 0
