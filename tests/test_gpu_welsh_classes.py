"""GPU parity of the class-specialised fused Welsh kernels (kernels.h, "Workgroup KINDS"): every
combination of oscillator-1 waveform x oscillator-2 waveform x (LFO waveform, LFO routing) x
(static | envelope-retuned filter) runs through the copy of the block body compiled for its
(LFO class, oscillator classes) triple and is compared with the f64 oracle.

Each patch fills one workgroup (256 identical voices), so the workgroup's class is exactly the
patch's; a bank holds 128 patches and the check is on its bus, which the oracle reproduces from one
voice per patch (x 256).  Tolerance: bus / V RMS <= 1e-6 (bar 1e-5).  8,100 patches in total (the five routings of LfoRoutingType and
the five more the shipped patch files use: pitch-osc2, pw-osc1, pw-osc2, resonance, cutoff-amp); the few
of them whose oracle output is unbounded (a noise LFO on the cutoff retunes the filter randomly every
frame; some depth / cutoff pairs grow to 1e10, chaotically, in oracle and device alike) stay idle."""
import itertools

import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["serial", "split", "time-parallel"])
def kernel_form(request, gpu_ctx):
    """Every test of this module runs against the three forms of the Welsh render: one voice per lane walking the frames
    (kernels.h), the same walk split over four (or three, or two) wavefronts per 64 voices (welsh_split.h: "split", the default for banks
    too big for the third form), and one wavefront per voice with the frames over its lanes (welsh_tp.h, the default
    for banks this small)."""
    old, old_split = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves
    gpu_ctx.time_parallel_max_voices = old if request.param == "time-parallel" else 0
    gpu_ctx.split_max_waves = 0 if request.param == "serial" else old_split
    yield request.param
    gpu_ctx.time_parallel_max_voices = old
    gpu_ctx.split_max_waves = old_split

WAVES = [T.WAVE_NONE, T.WAVE_SINE, T.WAVE_SQUARE, T.WAVE_PULSE_WIDTH, T.WAVE_TRIANGLE, T.WAVE_SAWTOOTH,
         T.WAVE_NOISE, T.WAVE_TRIANGLE_SINE, T.WAVE_DEBUG_MAX]
LFO_WAVES = [T.WAVE_SINE, T.WAVE_TRIANGLE, T.WAVE_SQUARE, T.WAVE_SAWTOOTH, T.WAVE_NOISE]
ROUTINGS = [T.LFO_NONE, T.LFO_AMPLITUDE, T.LFO_PITCH, T.LFO_PULSE_WIDTH, T.LFO_FILTER_CUTOFF,
            T.LFO_PITCH_OSC2, T.LFO_PW_OSC1, T.LFO_PW_OSC2, T.LFO_RESONANCE, T.LFO_CUTOFF_AMP]
CUTOFF_ROUTINGS = (T.LFO_FILTER_CUTOFF, T.LFO_CUTOFF_AMP)
PER_PATCH = 256   # one workgroup of four 64-lane waves
PER_BANK = 128


def _patch(k, w1, w2, wl, routing, retune):
    p = P.welsh_patch(k % P.N_PATCHES)  # envelopes, tunings, pan etc. from the synthetic table
    p.oscillator_1.waveform, p.oscillator_1.duty = w1, (0.3 if w1 == T.WAVE_PULSE_WIDTH else 0.5)
    p.oscillator_2.waveform, p.oscillator_2.duty = w2, (0.15 if w2 == T.WAVE_PULSE_WIDTH else 0.5)
    p.oscillator_mix = 0.6
    p.oscillator_2_sync = 1 if k % 3 == 0 else 0
    p.lfo_waveform, p.lfo_routing = wl, routing
    p.lfo_frequency = [0.53, 2.07, 5.13, 7.49][k % 4]
    p.lfo_depth = [0.05, 0.2, 0.5][k % 3]
    p.filter_cutoff_end = 0.5 if (retune and routing not in CUTOFF_ROUTINGS) else 0.0
    return p


def test_every_class_combination_against_oracle(gpu_ctx, oracle):
    from groove_amd import entities as E
    combos = list(itertools.product(WAVES, WAVES, LFO_WAVES, ROUTINGS, (False, True)))
    frames, blocks = 256, 3
    worst = 0.0
    for start in range(0, len(combos), PER_BANK):
        chunk = combos[start:start + PER_BANK]
        table = (T.WelshParams * len(chunk))(*[_patch(start + i, *c) for i, c in enumerate(chunk)])
        keys1 = (40 + (5 * (start + np.arange(len(chunk)))) % 37).astype(np.uint8)
        n = len(chunk) * PER_PATCH
        size = __import__("ctypes").sizeof(T.WelshParams)
        raw = np.frombuffer(bytes(bytearray(table)), dtype=np.uint8).reshape(len(chunk), size)
        params = (T.WelshParams * n).from_buffer_copy(np.repeat(raw, PER_PATCH, axis=0).tobytes())
        probe = oracle.Bank.welsh(table)  # which patches are numerically meaningful at all
        probe.note_events(T.note_events_np(np.arange(len(chunk), dtype=np.uint32), keys1, True))
        stable = np.abs(np.concatenate([probe.render(frames) for _ in range(blocks)], axis=1)).max(axis=(0, 1)) <= 4.0
        play = np.flatnonzero(stable).astype(np.uint32)
        voices = (play[:, None] * PER_PATCH + np.arange(PER_PATCH, dtype=np.uint32)[None, :]).ravel()
        synth = E.WelshSynth(gpu_ctx, params)
        synth.handle_midi_events(T.note_events_np(voices, np.repeat(keys1[play], PER_PATCH), True))
        ob = oracle.Bank.welsh(table)
        ob.note_events(T.note_events_np(play, keys1[play], True))
        bus = gpu_ctx.bus(blocks * frames)
        want = []
        for b in range(blocks):
            if b == 2:  # release: the envelope-retuned kinds move their cutoff again
                synth.handle_midi_events(T.note_events_np(voices, np.repeat(keys1[play], PER_PATCH), False))
                ob.note_events(T.note_events_np(play, keys1[play], False))
            synth.render_mix(bus, frames, at_frame=b * frames)
            want.append(ob.render(frames))  # [2][frames][patches]
        got = bus.download().astype(np.float64)
        want = np.concatenate(want, axis=1)
        want_bus = want.sum(axis=2).T * PER_PATCH  # [frames][2]
        assert np.isfinite(got).all()
        rms = np.sqrt(np.mean(((got - want_bus) / n) ** 2))
        worst = max(worst, rms)
        assert rms <= 1e-6, (start, rms)
        synth.destroy(); bus.destroy()
    assert worst > 0.0

