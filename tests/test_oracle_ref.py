"""CPU tier: the oracle's 24 dB low-pass and RBJ low-pass against the reference's own C text
(doc/filters004.txt, compiled from where it lies into oracle/_ref/libfilters004.so by
`make -C oracle ref`; the prebuilt .so travels to the GPU box, the source never enters the repo)."""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref_lib()
    if r is None:
        pytest.skip("oracle/_ref/libfilters004.so not built (reference tree absent)")
    return r


def _szx(ref, b1, b2, fc, fs=44100.0):
    k = C.c_double(1.0)
    coef = (C.c_float * 4)()
    ref.ref_szxform(1.0, 0.0, 0.0, 1.0, b1, b2, fc, fs, C.byref(k), coef)
    return k.value, list(coef)


@pytest.mark.parametrize("fc,ripple", [(1000.0, 0.8), (40.0, 0.707), (12000.0, 2.0), (5000.0, 1.3), (19000.0, 0.9)])
def test_lp24_sections_match_reference_bilinear(oracle, ref, fc, ripple):
    """Each section of the oracle's 24 dB low-pass is the prewarped bilinear transform of
    1/(c s^2 + d s + 1); filters004.txt's szxform() performs exactly that transform."""
    out = (C.c_double * 6)()
    oracle.lib().oracle_lp24_coeffs(fc, ripple, 44100.0, out)
    sg, cg = math.sinh(ripple), math.cosh(ripple) ** 2
    c0, c2 = 1 / (cg - 0.85355339059327376220), 1 / (cg - 0.14644660940672623780)
    for i, (c, d) in enumerate(((c0, c0 * sg * 1.84775906502257351226), (c2, c2 * sg * 0.76536686473017954346))):
        k, (beta1, beta2, alpha1, alpha2) = _szx(ref, d, c, fc)
        b0, a1, a2 = out[3 * i], out[3 * i + 1], out[3 * i + 2]
        assert abs(k - b0) <= 1e-14 * max(1.0, abs(b0))          # section gain (double in the C text)
        assert abs(-beta1 - a1) <= 2.5e-7 * max(1.0, abs(a1))    # coefficients are stored as float there
        assert abs(-beta2 - a2) <= 2.5e-7 * max(1.0, abs(a2))
        assert alpha1 == 2.0 and alpha2 == 1.0                   # numerator (1 + z^-1)^2


def test_rbj_lowpass_is_the_same_bilinear_transform(oracle, ref):
    """The cookbook LPF prototype H(s) = 1/(s^2 + s/Q + 1) through the reference's szxform()
    gives the oracle's normalised RBJ coefficients (b0 = gain, b1 = 2 b0, b2 = b0)."""
    out = (C.c_double * 5)()
    for f0, q in ((1000.0, 0.707), (250.0, 2.0), (8000.0, 0.5)):
        oracle.lib().oracle_rbj_lowpass(f0, q, 44100.0, out)
        k, (beta1, beta2, alpha1, alpha2) = _szx(ref, 1.0 / q, 1.0, f0)
        assert abs(k - out[0]) <= 1e-12
        assert abs(out[1] - 2 * out[0]) <= 1e-15 and abs(out[2] - out[0]) <= 1e-15
        assert abs(beta1 - out[3]) <= 2.5e-7 and abs(beta2 - out[4]) <= 2.5e-7


def test_lp24_recurrence_matches_reference_iir_filter(oracle, ref):
    """Run a sawtooth through the reference's cascaded direct-form-II iir_filter() (float
    state) with the reference's own coefficients, and through the oracle's transposed-DF-II
    f64 sections: same transfer function, so the outputs agree to float precision."""
    fs, fc, ripple, n = 44100.0, 1800.0, 0.9, 4096
    sg, cg = math.sinh(ripple), math.cosh(ripple) ** 2
    c0, c2 = 1 / (cg - 0.85355339059327376220), 1 / (cg - 0.14644660940672623780)
    coef = (C.c_float * 9)()
    gain = C.c_double(1.0)
    for i, (c, d) in enumerate(((c0, c0 * sg * 1.84775906502257351226), (c2, c2 * sg * 0.76536686473017954346))):
        sec = (C.c_float * 4)()
        ref.ref_szxform(1.0, 0.0, 0.0, 1.0, d, c, fc, fs, C.byref(gain), sec)
        for j in range(4):
            coef[1 + 4 * i + j] = sec[j]
    coef[0] = gain.value
    t = np.arange(n)
    x = (2.0 * ((t * 220.0 / fs) % 1.0) - 1.0).astype(np.float32)
    y_ref = np.zeros(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    ref.ref_iir_run(coef, 2, x.ctypes.data_as(fp), y_ref.ctypes.data_as(fp), n)
    y = np.zeros(n)
    dp = C.POINTER(C.c_double)
    xd = x.astype(np.float64)
    oracle.lib().oracle_lp24_run(fc, ripple, fs, xd.ctypes.data_as(dp), y.ctypes.data_as(dp), n)
    assert np.max(np.abs(y)) > 0.3
    assert np.max(np.abs(y - y_ref)) <= 2e-5  # float coefficients + float state on the reference side
