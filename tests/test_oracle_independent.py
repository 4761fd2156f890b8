"""CPU tier: the oracle against INDEPENDENT implementations of the published algorithms it claims.

Most rows of SURVEY.md section 8(a) are "parity unpinned": the reference holds no vector for them, so `oracle/` is a
restatement of a recollection and the golden file is its own output.  Nothing can turn that green.
What CAN be done is to make the restatement harder to doubt: every test here computes the same thing a second time by a
differently written route — scipy.signal's filter design and `lfilter`, closed forms in extended precision, exact integer
arithmetic — and asks the oracle to agree to f64 rounding.  A typo in one of the oracle's recurrences, a swapped
coefficient, an off-by-one delay length or a wrong constant fails here; a wrong RECOLLECTION of what Groove does cannot.

  a1  Oscillator        closed-form phase n f / SR in extended precision + waveform formulas; musicdsp noise in Python ints
  a2  Envelope          closed-form quadratic stages from the stage boundaries
  a3  BiQuad 12 dB      the cookbook's ANALOG prototypes through scipy.signal.bilinear (pre-warped), then lfilter
  a4  24 dB low-pass    scipy.signal.cheby1 (4th order, type I): analog poles, and the digital filter up to its DC gain
  a9  Bitcrusher        integer numpy
  a10 Chorus, a11 Delay, a12 Reverb   sparse-coefficient lfilter (combs, all-passes, taps)
  a13 Dca               the pan law at its anchor points
  a5  WelshVoice        the whole voice for static-filter patches, composed as array arithmetic (phases, scipy filter, envelopes, pan)
  a6  FM voice          running-sum carrier phase under the enveloped modulator;  a7 sampler: pointer stepping, exact
"""
import math

import numpy as np
import pytest
from scipy import signal

from groove_amd import abi_types as T

SR = 44100.0
TOL = 1e-12


def _dp(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _noise(n, seed=7):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


# ------------------------------------------------------------------------------------------ a3 BiQuad 12 dB
def _analog_prototype(kind, f0, q, bw_hz, db_gain):
    """(b, a) of the cookbook's s-domain prototype at unit cutoff (doc/Audio-EQ-Cookbook.txt: "H(s) = ..." of every mode),
    with the 1/Q each mode's digital `alpha` corresponds to (alpha = sin(w0) / (2 Q))."""
    w0 = 2.0 * math.pi * f0 / SR
    A = 10.0 ** (db_gain / 40.0)
    if kind in (T.FX_BIQUAD_BP12, T.FX_BIQUAD_BS12):
        lo, hi = f0 - bw_hz / 2.0, f0 + bw_hz / 2.0
        octaves = math.log2(hi / lo) if lo > 0.0 and hi / lo <= 256.0 else 8.0
        inv_q = 2.0 * math.sinh(math.log(2.0) / 2.0 * octaves * w0 / math.sin(w0))  # the cookbook's digital-compensated BW
    elif kind == T.FX_BIQUAD_PEAK12:
        inv_q = math.sqrt(2.0)                                                          # Q = 1 / sqrt 2 (DSP_SPEC)
    elif kind in (T.FX_BIQUAD_LSHELF12, T.FX_BIQUAD_HSHELF12):
        inv_q = math.sqrt((A + 1.0 / A) * (1.0 / 1.0 - 1.0) + 2.0)                      # shelf slope S = 1
    else:
        inv_q = 1.0 / q
    if kind == T.FX_BIQUAD_LP12:
        return [1.0], [1.0, inv_q, 1.0]
    if kind == T.FX_BIQUAD_HP12:
        return [1.0, 0.0, 0.0], [1.0, inv_q, 1.0]
    if kind == T.FX_BIQUAD_BP12:
        return [inv_q, 0.0], [1.0, inv_q, 1.0]
    if kind == T.FX_BIQUAD_BS12:
        return [1.0, 0.0, 1.0], [1.0, inv_q, 1.0]
    if kind == T.FX_BIQUAD_AP12:
        return [1.0, -inv_q, 1.0], [1.0, inv_q, 1.0]
    if kind == T.FX_BIQUAD_PEAK12:
        return [1.0, A * inv_q, 1.0], [1.0, inv_q / A, 1.0]
    sa = math.sqrt(A)
    if kind == T.FX_BIQUAD_LSHELF12:
        return [A, A * sa * inv_q, A * A], [A, sa * inv_q, 1.0]
    return [A * A, A * sa * inv_q, A], [1.0, sa * inv_q, A]  # high shelf


BIQUAD_KINDS = ["FX_BIQUAD_LP12", "FX_BIQUAD_HP12", "FX_BIQUAD_BP12", "FX_BIQUAD_BS12", "FX_BIQUAD_AP12", "FX_BIQUAD_PEAK12",
                "FX_BIQUAD_LSHELF12", "FX_BIQUAD_HSHELF12"]


@pytest.mark.parametrize("name", BIQUAD_KINDS)
@pytest.mark.parametrize("f0,q,bw,db", [(1000.0, 0.707, 500.0, 6.0), (120.0, 4.0, 60.0, -9.0), (9000.0, 0.5, 4000.0, 3.0)])
def test_rbj_modes_are_the_bilinear_transform_of_the_cookbooks_analog_prototypes(oracle, name, f0, q, bw, db):
    kind = getattr(T, name)
    b_s, a_s = _analog_prototype(kind, f0, q, bw, db)
    # pre-warping: the prototype's unit frequency lands on w0 when s is scaled by tan(w0 / 2) and the transform is (z-1)/(z+1)
    k = math.tan(math.pi * f0 / SR)
    nb, na = len(b_s), len(a_s)
    b_w = [c / k ** (nb - 1 - i) for i, c in enumerate(b_s)]
    a_w = [c / k ** (na - 1 - i) for i, c in enumerate(a_s)]
    bz, az = signal.bilinear(b_w, a_w, fs=0.5)
    bz, az = bz / az[0], az / az[0]
    p = T.fx_params(cutoff_hz=f0, q=q, bandwidth_hz=bw, db_gain=db)
    got = np.zeros(5)
    assert oracle.lib().oracle_rbj_for_kind(kind, p, SR, _dp(got)) == 1
    # the parameters travel as fp32: rebuild the expectation from what the oracle saw
    f0f, qf, bwf, dbf = (float(np.float32(v)) for v in (f0, q, bw, db))
    b_s, a_s = _analog_prototype(kind, f0f, qf, bwf, dbf)
    k = math.tan(math.pi * f0f / SR)
    bz, az = signal.bilinear([c / k ** (len(b_s) - 1 - i) for i, c in enumerate(b_s)], [c / k ** (len(a_s) - 1 - i) for i, c in enumerate(a_s)], fs=0.5)
    bz, az = bz / az[0], az / az[0]
    want = np.array([bz[0], bz[1], bz[2], az[1], az[2]])
    assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, np.max(np.abs(want))), (got, want)
    # Direct Form 1 run == lfilter with the oracle's own coefficients
    x = _noise(4096)
    y = np.zeros_like(x)
    oracle.lib().oracle_biquad_df1_run(_dp(got), _dp(x), _dp(y), len(x))
    ref = signal.lfilter(got[:3], np.concatenate([[1.0], got[3:]]), x)
    assert np.max(np.abs(y - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))


# ------------------------------------------------------------------------------------------ a4 24 dB low-pass
@pytest.mark.parametrize("ripple", [0.2, 0.707, 1.607, 2.2])  # (beyond ~3 the dB figure scipy takes, 10 log10(1 + eps^2), no longer resolves eps)
def test_lp24_sections_are_a_fourth_order_chebyshev_type_one(oracle, ripple):
    """The two s-plane sections the oracle transforms, 1 / (c s^2 + d s + 1) with the Appendix-A.4 constants, have exactly
    the poles of scipy's analog 4th-order Chebyshev type I filter whose ripple parameter is `ripple`
    (mu = asinh(1 / eps) / 4 = ripple)."""
    eps = 1.0 / math.sinh(4.0 * ripple)
    rp_db = 10.0 * math.log10(1.0 + eps * eps)
    _, poles, _ = signal.cheby1(4, rp_db, 1.0, analog=True, output="zpk")
    sg, cg = math.sinh(ripple), math.cosh(ripple) ** 2
    mine = []
    for sin2, two_sin in ((math.cos(math.pi / 8) ** 2, 2.0 * math.cos(math.pi / 8)), (math.sin(math.pi / 8) ** 2, 2.0 * math.sin(math.pi / 8))):
        c = 1.0 / (cg - sin2)
        mine.extend(np.roots([c, c * sg * two_sin, 1.0]))
    key = lambda z: (round(z.real, 9), round(z.imag, 9))
    assert np.allclose(sorted(mine, key=key), sorted(poles, key=key), rtol=1e-9, atol=1e-12)
    # and the constants the oracle spells out are those trigonometric values
    assert abs(math.cos(math.pi / 8) ** 2 - 0.85355339059327376220) < 1e-15 and abs(2 * math.cos(math.pi / 8) - 1.84775906502257351226) < 1e-15
    assert abs(math.sin(math.pi / 8) ** 2 - 0.14644660940672623780) < 1e-15 and abs(2 * math.sin(math.pi / 8) - 0.76536686473017954346) < 1e-15


@pytest.mark.parametrize("fc,ripple", [(1000.0, 0.707), (250.0, 1.607), (6000.0, 0.3), (40.0, 0.707)])
def test_lp24_run_is_scipys_digital_chebyshev_up_to_its_dc_gain(oracle, fc, ripple):
    """scipy.signal.cheby1(4, rp, fc, fs) — pre-warped bilinear transform, as the oracle — filters the same input to the
    same output times sqrt(1 + eps^2): an even-order Chebyshev has DC gain 1 / sqrt(1 + eps^2), the oracle's sections are
    each normalised to DC gain 1."""
    eps = 1.0 / math.sinh(4.0 * ripple)
    rp_db = 10.0 * math.log10(1.0 + eps * eps)
    sos = signal.cheby1(4, rp_db, fc, fs=SR, output="sos")
    x = _noise(6000, seed=11)
    y = np.zeros_like(x)
    oracle.lib().oracle_lp24_run(fc, ripple, SR, _dp(x), _dp(y), len(x))
    ref = signal.sosfilt(sos, x) * math.sqrt(1.0 + eps * eps)
    scale = max(1e-3, np.max(np.abs(ref)))
    assert np.max(np.abs(y - ref)) <= 1e-9 * scale, np.max(np.abs(y - ref)) / scale
    # the oracle's own six coefficients through lfilter, section by section (its transposed form II == any other form)
    c = np.zeros(6)
    oracle.lib().oracle_lp24_coeffs(fc, ripple, SR, _dp(c))
    z = x
    for b0, a1, a2 in (c[:3], c[3:]):  # the oracle carries a1, a2 with the feedback's sign folded in: y = ... + a1 y1 + a2 y2
        z = signal.lfilter([b0, 2 * b0, b0], [1.0, -a1, -a2], z)
    assert np.max(np.abs(y - z)) <= 1e-10 * scale


# ------------------------------------------------------------------------------------------ a10-a12 delay lines
def _fx_run(oracle, kind, x, **kw):
    fx = oracle.Fx(kind, (T.FxParams * 1)(T.fx_params(**kw)))
    blk = np.zeros((2, len(x), 1))
    blk[0, :, 0] = x
    blk[1, :, 0] = -0.5 * x
    fx.process(blk)
    return blk[0, :, 0], blk[1, :, 0]


def _delay_frames(seconds):
    return max(1, int(math.floor(float(np.float32(seconds)) * SR + 0.5)))


def test_delay_is_a_pure_delay(oracle):
    x = _noise(12000, seed=3)
    N = _delay_frames(0.1)
    yl, yr = _fx_run(oracle, T.FX_DELAY, x, delay_seconds=0.1)
    b = np.zeros(N + 1); b[N] = 1.0
    assert np.array_equal(yl, signal.lfilter(b, [1.0], x)) and np.array_equal(yr, signal.lfilter(b, [1.0], -0.5 * x))


def test_chorus_is_its_taps(oracle):
    x = _noise(30000, seed=4)
    N, voices = _delay_frames(0.25), 4
    spacing = N // voices
    yl, _ = _fx_run(oracle, T.FX_CHORUS, x, delay_seconds=0.25, voices=voices)
    b = np.zeros(N + 1)
    for k in range(voices):
        b[N - k * spacing] += 1.0
    assert np.max(np.abs(yl - signal.lfilter(b, [1.0], x))) <= TOL


def test_reverb_is_four_recirculating_combs_and_two_allpasses(oracle):
    x = _noise(20000, seed=5)
    att, seconds = 0.95, 1.25
    yl, yr = _fx_run(oracle, T.FX_REVERB, x, attenuation=att, reverb_seconds=seconds)

    def expect(inp):
        u = inp * float(np.float32(att))
        s = np.zeros_like(u)
        for d in (0.0297, 0.0371, 0.0411, 0.0437):      # out[n] = g (in[n - N] + out[n - N])
            N = max(1, int(math.floor(d * SR + 0.5)))
            g = 0.001 ** (d / float(np.float32(seconds)))
            b = np.zeros(N + 1); b[N] = g
            a = np.zeros(N + 1); a[0] = 1.0; a[N] = -g
            s += signal.lfilter(b, a, u)
        for d, dec in ((0.005, 0.09683), (0.0017, 0.03292)):  # H(z) = (z^-N - g) / (1 - g z^-N)
            N = max(1, int(math.floor(d * SR + 0.5)))
            g = 0.001 ** (d / dec)
            b = np.zeros(N + 1); b[0] = -g; b[N] = 1.0
            a = np.zeros(N + 1); a[0] = 1.0; a[N] = -g
            s = signal.lfilter(b, a, s)
        return s

    for got, inp in ((yl, x), (yr, -0.5 * x)):
        ref = expect(inp)
        assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


# ------------------------------------------------------------------------------------------ a1 Oscillator
def _osc(oracle, waveform, freq, n, duty=0.5, tune=1.0, fm=None):
    import ctypes as C
    p = T.OscillatorParams(waveform=waveform, duty=duty, tune=tune, fixed_hz=0.0)
    out = np.zeros(n)
    st = (C.c_uint32 * 2)()
    oracle.lib().oracle_oscillator_run(C.byref(p), freq, SR, _dp(fm) if fm is not None else None, _dp(out), n, st)
    return out, (st[0], st[1])


@pytest.mark.parametrize("freq", [440.0, 61.735, 5274.04])
def test_oscillator_is_a_phase_accumulator_with_the_published_waveforms(oracle, freq):
    n = 20000
    # position of frame i: frac(i f / SR) (the first tick emits position 0), here in extended precision
    pos = np.arange(n, dtype=np.longdouble) * (np.longdouble(freq) / np.longdouble(SR))
    pos = (pos - np.floor(pos)).astype(np.float64)
    away = lambda edge: np.abs(pos - edge) > 1e-7   # a frame within rounding of a waveform edge may legitimately fall on either side

    sine, _ = _osc(oracle, T.WAVE_SINE, freq, n)
    assert np.max(np.abs(sine - np.sin(2 * np.pi * pos))) <= 1e-8                      # (accumulated phase: n additions of f / SR)
    tri, _ = _osc(oracle, T.WAVE_TRIANGLE, freq, n)
    assert np.max(np.abs(tri - (4.0 * np.abs(pos - np.floor(pos + 0.5)) - 1.0))) <= 1e-8
    saw, _ = _osc(oracle, T.WAVE_SAWTOOTH, freq, n)
    ok = away(0.5)
    assert np.max(np.abs(saw - 2.0 * (pos - np.floor(pos + 0.5)))[ok]) <= 1e-8
    sq, _ = _osc(oracle, T.WAVE_SQUARE, freq, n)
    ok = away(0.5) & away(0.0) & away(1.0)
    assert np.array_equal(sq[ok], np.where(pos < 0.5, 1.0, -1.0)[ok])
    pw, _ = _osc(oracle, T.WAVE_PULSE_WIDTH, freq, n, duty=0.25)
    ok = away(0.25) & away(0.0) & away(1.0)
    assert np.array_equal(pw[ok], np.where(pos < 0.25, 1.0, -1.0)[ok])
    # frequency tune and 2^fm modulation scale the increment
    tuned, _ = _osc(oracle, T.WAVE_SINE, freq / 2.0, n, tune=2.0)
    assert np.max(np.abs(tuned - sine)) <= 1e-9
    fm = np.full(n, 1.0)
    up, _ = _osc(oracle, T.WAVE_SINE, freq / 2.0, n, fm=fm)
    assert np.max(np.abs(up - sine)) <= 1e-9


def test_noise_is_the_musicdsp_generator_in_exact_integers(oracle):
    n = 5000
    got, (s1, s2) = _osc(oracle, T.WAVE_NOISE, 440.0, n)
    x1, x2 = 0x70F4F854, 0xE1E9F0A7
    want = np.zeros(n)
    for i in range(n):
        x1 ^= x2
        want[i] = (x2 - (1 << 32) if x2 & 0x80000000 else x2) / 2147483648.0
        x2 = (x2 + x1) & 0xFFFFFFFF
    assert np.array_equal(got, want) and (s1, s2) == (x1, x2)


# ------------------------------------------------------------------------------------------ a2 Envelope
@pytest.mark.parametrize("a,d,s,r,off", [(0.01, 0.05, 0.6, 0.1, 6000), (0.0, 0.02, 0.25, 0.03, 3000), (0.05, 0.0, 1.0, 0.0, 4000),
                                          (0.2, 0.3, 0.5, 0.2, 2000)])
def test_envelope_is_four_closed_form_stages(oracle, a, d, s, r, off):
    """value = A + (B - A) (2t - t^2), t = frames into the stage / stage length; stage lengths scale with the distance to
    cover; zero-length stages are skipped in the same frame; a note-off releases from the current level."""
    import ctypes as C
    n = 12000
    p = T.EnvelopeParams(a, d, s, r)
    got = np.zeros(n)
    oracle.lib().oracle_envelope_run(C.byref(p), SR, off, _dp(got), n)
    af, df, sf, rf = a, d, s, r  # (envelope parameters travel as f64)
    want = np.zeros(n)

    def stage(start, A, B, length):
        """Frames start .. of a stage from A towards B over `length` frames; returns the frame after its last one."""
        N = int(math.ceil(length * (1.0 - 2.0 ** -16))) if length > 0.0 else 0
        i = np.arange(N)
        t = i / length if N else i
        seg = A + (B - A) * (2.0 * t - t * t)
        end = min(n, start + N)
        want[start:end] = seg[: end - start]
        return start + N

    f = stage(0, 0.0, 1.0, af * SR)                 # attack
    if f < off:
        f = stage(f, 1.0, sf, df * SR * (1.0 - sf))  # decay
    want[min(f, n):off] = sf                         # sustain
    want[off:] = 0.0
    level = want[off - 1] if off > 0 else 0.0        # release from the level of the last tick
    stage(off, level, 0.0, rf * SR * level)
    assert np.max(np.abs(got - want)) <= TOL


# ------------------------------------------------------------------------------------------ a9 Bitcrusher, a13 Dca
def test_bitcrusher_is_an_integer_quantise(oracle):
    xs = np.concatenate([_noise(4000, seed=9) * 1.2, [0.0, 1.0, -1.0, 0.5, 3e-5]]).astype(np.float32)
    for bits in (0, 1, 5, 8, 13, 15):
        got = np.array([oracle.lib().oracle_bitcrush_f32(float(x), bits) for x in xs], dtype=np.float32)
        q = (np.abs(xs) * np.float32(32767.0)).astype(np.uint32)           # truncation on the 16-bit scale
        q = (q >> np.uint32(bits)) << np.uint32(bits)
        want = np.copysign(q.astype(np.float32) * np.float32(1.0 / 32767.0), xs)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), bits


def test_dca_pan_law_anchor_points(oracle):
    lr = np.zeros(2)
    L = oracle.lib()
    L.oracle_dca(1.0, 1.0, 0.0, _dp(lr)); assert np.allclose(lr, [0.75, 0.75], atol=1e-15)   # centre
    L.oracle_dca(1.0, 1.0, -1.0, _dp(lr)); assert np.allclose(lr, [1.0, 0.0], atol=1e-15)    # hard left
    L.oracle_dca(1.0, 1.0, 1.0, _dp(lr)); assert np.allclose(lr, [0.0, 1.0], atol=1e-15)     # hard right
    L.oracle_dca(0.5, 0.5, 0.3, _dp(lr))
    assert np.allclose(lr, [0.25 * (1 - 0.25 * 1.3 ** 2), 0.25 * (1 - (0.15 - 0.5) ** 2)], atol=1e-15)


# ------------------------------------------------------------------------------------------ a5 WelshVoice, composed independently
def _closed_form_envelope(a, d, s, r, off, n):
    """The four quadratic stages (docs/DSP_SPEC.md section 3) with note-on at frame 0 and note-off at frame `off`; returns the
    values and the first frame at which the envelope is idle again."""
    v = np.zeros(n)

    def stage(start, A, B, length):
        N = int(math.ceil(length * (1.0 - 2.0 ** -16))) if length > 0.0 else 0
        i = np.arange(N)
        t = i / length if N else i
        end = min(n, start + N)
        v[start:end] = (A + (B - A) * (2.0 * t - t * t))[: end - start]
        return start + N

    f = stage(0, 0.0, 1.0, a * SR)
    if f < off:
        f = stage(f, 1.0, s, d * SR * (1.0 - s))
    v[min(f, n):off] = s
    v[off:] = 0.0
    level = v[off - 1]
    idle_from = stage(off, level, 0.0, r * SR * level)
    return v, idle_from


def _lp24_sections(ripple):
    """(c, d) of the two analog sections 1 / (c s^2 + d s + 1): the 4th-order Chebyshev pole pairs (docs/DSP_SPEC.md section 4; the
    tests above show them equal to scipy's cheb1ap up to the DC gain)."""
    cg, sg = math.cosh(ripple) ** 2, math.sinh(ripple)
    half = math.sqrt(0.5)
    out = []
    for cr, dk in ((0.5 + 0.5 * half, 2.0 * math.cos(math.pi / 8)), (0.5 - 0.5 * half, 2.0 * math.cos(3 * math.pi / 8))):  # 0.853553.., 1.847759..; 0.146446.., 0.765366..
        c = 1.0 / (cg - cr)
        out.append((c, c * sg * dk))
    return out


def _lp24_time_varying(x_in, fc, ripple):
    """The 24 dB low-pass with the cutoff of EVERY frame given (fc[j], Hz, clamped to [1 Hz, 0.49 SR]): each of the two sections is
    scipy's bilinear transform of its analog prototype at k = tan(pi fc / SR), recomputed for the frame BEFORE the frame's sample
    goes through it, in transposed direct form II (the state carries what the OLD coefficients left in it)."""
    fc = np.clip(fc, 1.0, 0.49 * SR)
    sections = _lp24_sections(ripple)
    n = len(x_in)
    y = np.empty(n)
    z = [[0.0, 0.0], [0.0, 0.0]]
    coef = [None, None]
    last_fc = None
    for j in range(n):
        if fc[j] != last_fc:
            k = math.tan(math.pi * fc[j] / SR)
            for q, (c, d) in enumerate(sections):
                b, a = signal.bilinear([1.0], [c / (k * k), d / k, 1.0], fs=0.5)
                coef[q] = (b / a[0], a / a[0])
            last_fc = fc[j]
        x = x_in[j]
        for q in range(2):
            b, a = coef[q]
            out = b[0] * x + z[q][0]
            z[q][0] = b[1] * x - a[1] * out + z[q][1]
            z[q][1] = b[2] * x - a[2] * out
            x = out
        y[j] = x
    return y


def _waveform(w, pos, duty=0.5, noise=None):
    """docs/DSP_SPEC.md section 2, value() at position pos in [0, 1)."""
    if w == T.WAVE_SINE:
        return np.sin(2.0 * np.pi * pos)
    if w == T.WAVE_SQUARE or w == T.WAVE_PULSE_WIDTH:
        return np.where(pos < duty, 1.0, -1.0)
    if w == T.WAVE_TRIANGLE:
        return 4.0 * np.abs(pos - np.floor(pos + 0.5)) - 1.0
    if w == T.WAVE_SAWTOOTH:
        return 2.0 * (pos - np.floor(pos + 0.5))
    if w == T.WAVE_TRIANGLE_SINE:
        return 4.0 * np.abs(pos - np.floor(pos + 0.75) + 0.25) - 1.0
    if w == T.WAVE_NOISE:
        return noise
    return np.zeros(len(pos))


def _musicdsp_noise(n):
    x1, x2 = 0x70F4F854, 0xE1E9F0A7
    v = np.zeros(n)
    for i in range(n):
        x1 ^= x2
        v[i] = (x2 - (1 << 32) if x2 & 0x80000000 else x2) / 2147483648.0
        x2 = (x2 + x1) & 0xFFFFFFFF
    return v


def _independent_welsh_voice(p, key, n, off):
    """A Welsh voice written from the published pieces without one line of oracle/ (docs/DSP_SPEC.md section 6 is what is under
    test): oscillator pair (every waveform; hard sync; osc 2 fixed or tuned) under an LFO (sine / triangle / square / sawtooth) routed
    to nothing, the amplitude, the pitch, the pulse width or the filter cutoff; mix; the 24 dB low-pass — static, or RETUNED every
    frame by the filter envelope or the LFO (_lp24_time_varying: scipy's bilinear transform per frame); amplitude envelope; pan.
    Closed-form phases in extended precision.  Returns ([2][n], first idle frame)."""
    f_note = 440.0 * 2.0 ** ((key - 69) / 12.0)
    i = np.arange(n, dtype=np.longdouble)
    lpos = i * (np.longdouble(p.lfo_frequency) / np.longdouble(SR))    # (the LFO's first tick emits phase 0 too)
    lfo = _waveform(p.lfo_waveform, (lpos - np.floor(lpos)).astype(np.float64))
    depth = float(np.float32(p.lfo_depth))
    routing = p.lfo_routing
    ld = lfo * depth

    def positions(o):
        f = o.fixed_hz if o.fixed_hz > 0.0 else f_note * o.tune
        if routing == T.LFO_PITCH:   # the increment of frame j is f 2^(lfo_j depth) / SR; frame 0 emits phase 0
            inc = (np.longdouble(f) / np.longdouble(SR)) * np.exp2(ld.astype(np.longdouble))
            return np.concatenate([[np.longdouble(0.0)], np.cumsum(inc[1:])])
        return i * (np.longdouble(f) / np.longdouble(SR))

    def duty_of(o):
        d = 0.5 if o.waveform == T.WAVE_SQUARE else float(np.float32(o.duty))
        return np.clip(d * (1.0 + ld), 0.0, 1.0) if routing == T.LFO_PULSE_WIDTH else d

    noise = _musicdsp_noise(n) if T.WAVE_NOISE in (p.oscillator_1.waveform, p.oscillator_2.waveform) else None
    pos1 = positions(p.oscillator_1)
    wraps = np.floor(pos1)
    v1 = _waveform(p.oscillator_1.waveform, (pos1 - wraps).astype(np.float64), duty_of(p.oscillator_1), noise)
    pos2 = positions(p.oscillator_2)
    if p.oscillator_2_sync:          # osc 2 restarts at 0 on every frame at which osc 1 wrapped (constant increments here)
        assert routing != T.LFO_PITCH
        wrapped = np.concatenate([[False], wraps[1:] != wraps[:-1]])
        last = np.maximum.accumulate(np.where(wrapped, np.arange(n), 0))
        d2 = pos2[1] - pos2[0]
        pos2 = (i - last.astype(np.longdouble)) * d2
    v2 = _waveform(p.oscillator_2.waveform, (pos2 - np.floor(pos2)).astype(np.float64), duty_of(p.oscillator_2), noise)
    mix = float(np.float32(p.oscillator_mix))
    s = v1 * mix + v2 * (1.0 - mix)

    ripple = float(np.float32(p.filter_passband_ripple))
    start, end = float(np.float32(p.filter_cutoff_start)), float(np.float32(p.filter_cutoff_end))
    if end != 0.0:                   # the filter envelope retunes the filter every frame
        fe = p.filter_envelope
        env, _ = _closed_form_envelope(fe.attack, fe.decay, fe.sustain, fe.release, off, n)
        fc = 25.0 * 800.0 ** np.clip(start + (1.0 - start) * end * env, 0.0, 1.0)
    elif routing == T.LFO_FILTER_CUTOFF:
        fc = 25.0 * 800.0 ** np.clip(start * (1.0 + ld), 0.0, 1.0)
    else:
        fc = None
    if fc is not None:
        y = _lp24_time_varying(s, fc, ripple)
    elif ripple <= 2.2:              # static: scipy's own 4th-order Chebyshev design, up to its DC gain (beyond ~3 the dB figure no longer resolves eps)
        eps = 1.0 / math.sinh(4.0 * ripple)
        sos = signal.cheby1(4, 10.0 * math.log10(1.0 + eps * eps), float(np.float32(p.filter_cutoff_hz)), fs=SR, output="sos")
        y = signal.sosfilt(sos, s) * math.sqrt(1.0 + eps * eps)
    else:
        y = _lp24_time_varying(s, np.full(n, float(np.float32(p.filter_cutoff_hz))), ripple)

    e = p.amp_envelope
    amp, idle_from = _closed_form_envelope(e.attack, e.decay, e.sustain, e.release, off, n)
    if routing == T.LFO_AMPLITUDE:
        amp = amp * (1.0 + ld)
    m = y * amp * float(np.float32(p.dca_gain))
    pan = float(np.float32(p.dca_pan))
    out = np.stack([m * (1.0 - 0.25 * (pan + 1.0) ** 2), m * (1.0 - (0.5 * pan - 0.5) ** 2)])
    out[:, min(idle_from, n):] = 0.0
    return out, idle_from


def _welsh_patch(w1, w2, tune2, mix, env, cutoff, ripple, pan, lfo=None, duty=0.3, fixed2=0.0, routing=None, sweep=None):
    p = T.WelshParams()
    p.oscillator_1 = T.OscillatorParams(w1, duty, 1.0, 0.0)
    p.oscillator_2 = T.OscillatorParams(w2, duty, tune2, fixed2)
    p.oscillator_2_sync, p.oscillator_mix = 0, mix
    p.amp_envelope = T.EnvelopeParams(*env)
    p.filter_envelope = T.EnvelopeParams(0.01, 0.1, 0.5, 0.1)    # (runs, but drives nothing: filter_cutoff_end = 0)
    p.lfo_waveform, p.lfo_routing, p.lfo_frequency, p.lfo_depth = T.WAVE_SINE, (routing if routing is not None else (T.LFO_AMPLITUDE if lfo else T.LFO_NONE)), (lfo or (1.0, 0.0))[0], (lfo or (1.0, 0.0))[1]
    p.filter_cutoff_hz, p.filter_passband_ripple, p.filter_cutoff_start, p.filter_cutoff_end = cutoff, ripple, 0.5, 0.0
    if sweep:   # (cutoff percent start, end, filter envelope): the envelope retunes the filter every frame
        p.filter_cutoff_start, p.filter_cutoff_end, p.filter_envelope = sweep[0], sweep[1], T.EnvelopeParams(*sweep[2])
    p.dca_gain, p.dca_pan = 0.8, pan
    return p


def test_welsh_voice_composition_against_an_independent_array_implementation(oracle):
    """Seven hand-built patches x two keys (the last one with the LFO on the pitch of both oscillators): oscillator pair -> mix -> static 24 dB low-pass -> amplitude envelope (x LFO) -> Dca, with
    the note-off inside the render and the idle tail after the release.  Bar: 1e-8 of full scale (n additions of the phase
    increment against a closed form; a sample whose phase is within rounding of a waveform edge is skipped)."""
    patches = [
        _welsh_patch(T.WAVE_SAWTOOTH, T.WAVE_SINE, 2.0 ** (7 / 12), 0.6, (0.01, 0.05, 0.6, 0.08), 1200.0, 0.9, -0.4),
        _welsh_patch(T.WAVE_SINE, T.WAVE_TRIANGLE, 2.0, 0.5, (0.0, 0.02, 0.3, 0.05), 400.0, 0.707, 0.25, lfo=(5.13, 0.3)),
        _welsh_patch(T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH, 1.0, 0.25, (0.03, 0.0, 1.0, 0.02), 3000.0, 1.607, 1.0),
        _welsh_patch(T.WAVE_SINE, T.WAVE_SINE, 2.0 ** (5 / 1200), 0.5, (0.002, 0.3, 0.0, 0.3), 800.0, 1.2, 0.0, lfo=(0.53, 0.5)),
        _welsh_patch(T.WAVE_TRIANGLE, T.WAVE_NONE, 1.0, 1.0, (0.05, 0.1, 0.8, 0.1), 150.0, 0.8, -1.0),
        _welsh_patch(T.WAVE_SINE, T.WAVE_SAWTOOTH, 1.0, 0.7, (0.01, 0.05, 0.5, 0.04), 2000.0, 0.707, 0.5, fixed2=261.6255653),
        _welsh_patch(T.WAVE_SINE, T.WAVE_TRIANGLE, 2.0 ** (12 / 12), 0.5, (0.005, 0.1, 0.7, 0.06), 1500.0, 0.9, -0.2, lfo=(5.13, 0.05), routing=T.LFO_PITCH),  # vibrato
    ]
    n, off = 12000, 5000
    for key in (45, 72):
        params = (T.WelshParams * len(patches))(*patches)
        bank = oracle.Bank.welsh(params)
        lanes = np.arange(len(patches), dtype=np.uint32)
        bank.note_events(T.note_events_np(lanes, np.full(len(patches), key, dtype=np.uint8), True))
        got = bank.render(off)
        bank.note_events(T.note_events_np(lanes, np.full(len(patches), key, dtype=np.uint8), False))
        got = np.concatenate([got, bank.render(n - off)], axis=1)
        for k, p in enumerate(patches):
            want, idle_from = _independent_welsh_voice(p, key, n, off)
            assert idle_from < n, "the render must reach the idle tail"
            err = np.abs(got[:, :, k] - want)
            # saw / triangle: frames within rounding of the waveform's discontinuity may land on either side in either implementation
            suspicious = err.max(axis=0) > 1e-8
            assert suspicious.sum() <= 2 and err[:, ~suspicious].max() <= 1e-8, (key, k, err.max(), int(suspicious.sum()))
            assert np.abs(want).max() > 1e-3


RETUNED_PATCHES = lambda: [  # noqa: E731
    _welsh_patch(T.WAVE_SAWTOOTH, T.WAVE_SINE, 2.0 ** (7 / 12), 0.6, (0.01, 0.05, 0.6, 0.08), 1200.0, 0.9, -0.4, sweep=(0.2, 0.6, (0.02, 0.08, 0.4, 0.06))),
    _welsh_patch(T.WAVE_SINE, T.WAVE_TRIANGLE, 2.0, 0.5, (0.0, 0.02, 0.3, 0.05), 400.0, 0.707, 0.25, lfo=(5.13, 0.3), sweep=(0.45, 0.9, (0.0, 0.05, 0.2, 0.03))),
    _welsh_patch(T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH, 1.0, 0.25, (0.03, 0.0, 1.0, 0.02), 3000.0, 1.607, 1.0, sweep=(0.7, 0.3, (0.05, 0.0, 1.0, 0.02))),
]


def test_retuned_welsh_voice_against_an_independent_time_varying_filter(oracle):
    """The kinds three voices in four of the benchmark belong to: the filter envelope retunes the 24 dB low-pass every frame.
    Independent side: closed-form envelope -> cutoff law -> scipy's bilinear transform of each analog section per frame -> a
    transposed-direct-form-II loop written here.  Bar: 1e-8 of full scale (f64 on both sides)."""
    patches = RETUNED_PATCHES()
    n, off, key = 6000, 2500, 52
    params = (T.WelshParams * len(patches))(*patches)
    bank = oracle.Bank.welsh(params)
    lanes = np.arange(len(patches), dtype=np.uint32)
    bank.note_events(T.note_events_np(lanes, np.full(len(patches), key, dtype=np.uint8), True))
    got = bank.render(off)
    bank.note_events(T.note_events_np(lanes, np.full(len(patches), key, dtype=np.uint8), False))
    got = np.concatenate([got, bank.render(n - off)], axis=1)
    for k, p in enumerate(patches):
        want, idle_from = _independent_welsh_voice(p, key, n, off)
        assert idle_from < n and np.abs(want).max() > 1e-3
        err = np.abs(got[:, :, k] - want)
        suspicious = err.max(axis=0) > 1e-8
        assert suspicious.sum() <= 2 and err[:, ~suspicious].max() <= 1e-8, (k, err.max(), int(suspicious.sum()))


def BENCHMARK_KEYS():
    """The key the benchmark gives each patch's first voice (36 + 7 w mod 49) — except that an A (220 Hz x 2^k: a rational
    frequency / SR pair) moves up a semitone: at 220 Hz a waveform edge lands EXACTLY on frame 2,205, a tie that accumulated f64
    rounding decides in the oracle, exact integer arithmetic on the device and extended precision here (docs/DSP_SPEC.md section 2),
    and one flipped square-wave sample rings through the filter for a hundred frames."""
    from groove_amd import patches as P
    keys = (36 + (7 * np.arange(P.N_PATCHES)) % 49).astype(np.uint8)
    keys[keys % 12 == 9] += 1
    return keys


def test_every_synthetic_benchmark_patch_against_the_independent_voice(oracle):
    """All 32 synthetic Welsh patches of the benchmark projects (groove_amd/patches.py: every waveform incl. noise and triangle-sine,
    hard sync, fixed-frequency osc 2, LFOs of four waveforms routed to nothing / amplitude / pitch / pulse width / filter cutoff,
    static and envelope-retuned filters, ripples up to 3.2), each at the key the benchmark gives its first voice (BENCHMARK_KEYS),
    through note-on, note-off and the release: the oracle against _independent_welsh_voice.  1e-8 of full scale; a frame whose phase is within
    rounding of a waveform edge may fall on either side in either implementation (skipped, at most a handful)."""
    from groove_amd import patches as P
    n, off = 4000, 1800
    table = [P.welsh_patch(j) for j in range(P.N_PATCHES)]
    keys = BENCHMARK_KEYS()
    bank = oracle.Bank.welsh((T.WelshParams * len(table))(*table))
    lanes = np.arange(len(table), dtype=np.uint32)
    bank.note_events(T.note_events_np(lanes, keys, True))
    got = bank.render(off)
    bank.note_events(T.note_events_np(lanes, keys, False))
    got = np.concatenate([got, bank.render(n - off)], axis=1)
    for k, p in enumerate(table):
        want, _ = _independent_welsh_voice(p, int(keys[k]), n, off)
        assert np.abs(want).max() > 1e-3, k
        err = np.abs(got[:, :, k] - want)
        suspicious = err.max(axis=0) > 1e-8
        assert suspicious.sum() <= 2 and err[:, ~suspicious].max() <= 1e-8, (k, float(err.max()), int(suspicious.sum()), np.flatnonzero(suspicious)[:8])


# ------------------------------------------------------------------------------------------ a6 FM voice, a7 sampler
FM_CASES = ((2.0, 1.0, 1.0, 0.0), (3.5, 0.7, 10.0, -0.6), (0.5, 1.0, 0.1, 0.8))   # ratio, depth, beta, pan
FM_CARRIER_ENV, FM_MODULATOR_ENV, FM_GAIN = (0.01, 0.05, 0.7, 0.05), (0.0, 0.2, 0.4, 0.02), 0.9


def _fm_params(ratio, depth, beta, pan):
    return T.FmParams(ratio, depth, beta, T.EnvelopeParams(*FM_CARRIER_ENV), T.EnvelopeParams(*FM_MODULATOR_ENV), FM_GAIN, pan)


def _independent_fm_voice(ratio, depth, beta, pan, key, n, off):
    return _independent_fm_voice_of(_fm_params(ratio, depth, beta, pan), key, n, off)


def _independent_fm_voice_of(p, key, n, off):
    """Carrier phase = running sum of f_c / SR (1 + modulator x modulator envelope x depth x beta) — the first tick emits phase 0 —,
    modulator at f_c x ratio, output = sin(carrier) x carrier envelope -> Dca; extended precision, nothing of
    oracle/.  Returns ([2][n], first idle frame)."""
    fc = 440.0 * 2.0 ** ((key - 69) / 12.0)
    i = np.arange(n, dtype=np.longdouble)
    mpos = i * (np.longdouble(fc * p.ratio) / np.longdouble(SR))
    mod = np.sin(2.0 * np.pi * (mpos - np.floor(mpos)).astype(np.float64))
    ce, me = p.carrier_envelope, p.modulator_envelope
    cenv, idle_from = _closed_form_envelope(ce.attack, ce.decay, ce.sustain, ce.release, off, n)
    menv, _ = _closed_form_envelope(me.attack, me.decay, me.sustain, me.release, off, n)
    lfm = mod * menv * float(np.float32(p.depth)) * float(np.float32(p.beta))
    delta = (np.longdouble(fc) / np.longdouble(SR)) * (1.0 + lfm.astype(np.longdouble))
    cpos = np.concatenate([[np.longdouble(0.0)], np.cumsum(delta[1:])])
    car = np.sin(2.0 * np.pi * (cpos - np.floor(cpos)).astype(np.float64))
    m = car * cenv * float(np.float32(p.dca_gain))
    pf = float(np.float32(p.dca_pan))
    want = np.stack([m * (1.0 - 0.25 * (pf + 1.0) ** 2), m * (1.0 - (0.5 * pf - 0.5) ** 2)])
    want[:, min(idle_from, n):] = 0.0   # (the oscillators stop ticking while the carrier envelope is idle; nothing sounds there either way)
    return want, idle_from


def test_every_synthetic_fm_patch_against_the_independent_voice(oracle):
    """The 16 FM patches of config #5 (modulation indices 0.1 - 15), at the benchmark's keys, through note-off and release."""
    from groove_amd import patches as P
    n, off = 6000, 2500
    table = [P.fm_patch(j) for j in range(16)]
    keys = (36 + (7 * np.arange(16)) % 49).astype(np.uint8)
    bank = oracle.Bank.fm((T.FmParams * 16)(*table))
    lanes = np.arange(16, dtype=np.uint32)
    bank.note_events(T.note_events_np(lanes, keys, True))
    got = bank.render(off)
    bank.note_events(T.note_events_np(lanes, keys, False))
    got = np.concatenate([got, bank.render(n - off)], axis=1)
    for k, p in enumerate(table):
        want, _ = _independent_fm_voice_of(p, int(keys[k]), n, off)
        assert np.abs(want).max() > 0.05 and np.abs(got[:, :, k] - want).max() <= 5e-8, (k, float(np.abs(got[:, :, k] - want).max()))


def test_fm_voice_is_a_phase_modulated_sine(oracle):
    n, off, key = 9000, 4000, 57
    for ratio, depth, beta, pan in FM_CASES:
        bank = oracle.Bank.fm((T.FmParams * 1)(_fm_params(ratio, depth, beta, pan)))
        ev = lambda on: T.note_events_np(np.zeros(1, dtype=np.uint32), np.full(1, key, dtype=np.uint8), on)  # noqa: E731
        bank.note_events(ev(True))
        got = bank.render(off)
        bank.note_events(ev(False))
        got = np.concatenate([got, bank.render(n - off)], axis=1)[:, :, 0]
        want, idle_from = _independent_fm_voice(ratio, depth, beta, pan, key, n, off)
        assert idle_from < n and np.abs(want).max() > 0.3
        assert np.abs(got - want).max() <= 2e-8, (ratio, np.abs(got - want).max())


def test_sampler_is_pointer_stepping_without_interpolation(oracle):
    """out[i] = pcm[floor(i x step)] x gain until the pointer runs off the end; step = note frequency / root frequency, or 1 for a
    drumkit buffer (root 0); mono duplicated to both channels."""
    rng = np.random.default_rng(12)
    pcm = rng.uniform(-1, 1, 3000).astype(np.float32)
    descs = (T.SampleDesc * 2)(T.SampleDesc(0, 2000, 440.0), T.SampleDesc(2000, 1000, 0.0))
    params = (T.SamplerParams * 2)(T.SamplerParams(0, 1, 0.5), T.SamplerParams(1, 1, 1.0))
    for key in (69, 76, 60):
        bank = oracle.Bank.sampler(pcm, descs, params)
        bank.note_events(T.note_events_np(np.arange(2, dtype=np.uint32), np.full(2, key, dtype=np.uint8), True))
        n = 3500
        got = bank.render(n)
        for lane, (off0, length, root, gain) in enumerate(((0, 2000, 440.0, 0.5), (2000, 1000, 0.0, 1.0))):
            step = (440.0 * 2.0 ** ((key - 69) / 12.0)) / root if root > 0 else 1.0
            pos = np.arange(n) * step
            idx = np.floor(pos).astype(np.int64)
            want = np.where(idx < length, pcm[off0 + np.minimum(idx, length - 1)].astype(np.float64) * gain, 0.0)
            safe = np.abs(pos - np.round(pos)) > 1e-6 if step != 1.0 and key != 69 else np.ones(n, dtype=bool)  # (an accumulated pointer within rounding of an integer)
            assert np.array_equal(got[0, safe, lane], want[safe]) and np.array_equal(got[0, :, lane], got[1, :, lane]), (key, lane)
