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


def test_random_patches_every_kernel_form_against_the_oracle(gpu_ctx, oracle):
    """Patches DRAWN from a seed (groove_amd.patches.random_welsh_patch: every continuous parameter, every routing, instant attacks, zero
    sustains, cutoffs from 40 Hz to 20 kHz) instead of the 32 benchmark ones, eight per bank in runs of eight voices on random keys, 40 blocks
    with a note-off: every kernel form — time-parallel, the all-kinds serial kernel, role-split, the per-kind kernels with their fp32 filter
    bodies — against the f64 oracle voice by voice (<= 1e-5 RMS; measured worst of 60 x 64 voices: 2.1e-6), role-split bit for bit the
    serial kernel's.  No key is an A: 55 and 110 Hz are rational in 44,100 and put a square's edge EXACTLY on a frame, which the oracle's
    accumulated f64 phase and the device's 64-bit counter decide differently (docs/DSP_SPEC.md section 2)."""
    import os
    from groove_amd import entities as E
    n, blocks, off_at = 64, 40, 24
    old = (gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves)
    lanes = np.arange(n, dtype=np.uint32)
    try:
        for seed in range(int(os.environ.get("GROOVE_TEST_SEEDS", "6"))):   # (60 seeds ran clean at the end of round 5)
            rng = np.random.default_rng(seed)
            patches = [P.random_welsh_patch(rng) for _ in range(8)]
            params = (T.WelshParams * n)(*[patches[(i // 8) % 8] for i in range(n)])
            keys = rng.integers(30, 96, size=n).astype(np.uint8)
            keys[keys % 12 == 9] += 1
            ob = oracle.Bank.welsh(params)
            ob.note_events(T.note_events_np(lanes, keys, True))
            want = []
            for b in range(blocks):
                if b == off_at:
                    ob.note_events(T.note_events_np(lanes, keys, False))
                want.append(ob.render(256))
            want = np.concatenate(want, axis=1)
            assert np.sqrt(np.mean(want ** 2)) > 1e-2
            got = {}
            for form in ("tp", "any", "split", "per-kind"):
                gpu_ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                gpu_ctx.split_max_waves = (1 << 20) if form == "split" else 0
                gpu_ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
                s = E.WelshSynth(gpu_ctx, params)
                blk = gpu_ctx.block(n, 256)
                s.handle_midi_events(T.note_events_np(lanes, keys, True))
                out = []
                for b in range(blocks):
                    if b == off_at:
                        s.handle_midi_events(T.note_events_np(lanes, keys, False))
                    s.generate_batch_values(blk, 256)
                    out.append(blk.download(256))
                got[form] = np.concatenate(out, axis=1)
                rms = np.sqrt(np.mean((got[form].astype(np.float64) - want) ** 2, axis=(0, 1)))
                assert np.isfinite(got[form]).all() and rms.max() <= 1e-5, (seed, form, int(np.argmax(rms)), float(rms.max()))
                s.destroy(); blk.destroy()
            assert np.array_equal(got["split"].view(np.uint32), got["any"].view(np.uint32)), seed
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_random_note_event_sequences_in_every_kernel_form(gpu_ctx, oracle):
    """Note events drawn from a seed: per block up to a dozen events in ARBITRARY order — the same voice several times in one block
    (on, off, on again: the order decides), events for ALL voices, note-offs for idle voices, re-triggers of sounding ones — over ragged
    blocks, the 32 benchmark patches on 96 voices.  HandlesMidi's semantics (block-granular, applied in order at the next block's start) in
    every kernel form against the oracle voice by voice (<= 1e-5 RMS).
    One knife edge is allowed for, at most one voice of a seed and <= 2e-4: a re-trigger takes the envelope's level as the start of a new
    attack of `N = ceil(len)` frames, `len = attack x SR x (1 - level)`; the device's level is an fp32 polynomial, the oracle's f64, and
    when `len` comes within 3e-4 of an integer the two stages differ by ONE frame — the decay behind starts a frame apart, 1.5e-4 of level
    for a 0.3 s filter decay (seed 19, voice 3; docs/DSP_SPEC.md section 3; gone when the patch's attack is 0.06002 s instead of 0.06).
    The four forms agree with each other bit for bit in that voice too."""
    import os
    from groove_amd import entities as E
    n, blocks = 96, 36
    params = P.welsh_voices(n)
    old = (gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves)
    try:
        for seed in range(int(os.environ.get("GROOVE_TEST_SEEDS", "4"))):   # (100 seeds ran clean at the end of round 5)
            rng = np.random.default_rng(seed)
            script, sizes = [], []
            for b in range(blocks):
                evs = []
                for _ in range(int(rng.integers(0, 13)) if b else 0):
                    voice = T.ALL_VOICES if rng.random() < 0.06 else int(rng.integers(n)) if rng.random() < 0.7 else int(rng.integers(4))
                    key = int(rng.integers(30, 96))
                    evs.append((voice, key + (key % 12 == 9), bool(rng.random() < 0.65)))     # (no A: docs/DSP_SPEC.md section 2, ties)
                if b == 0:
                    evs = [(v, 36 + (7 * v) % 49, True) for v in range(0, n, 2)]
                script.append(evs)
                sizes.append(int(rng.choice([256, 256, 256, 100, 37, 1])))
            ob = oracle.Bank.welsh(params)
            want = []
            for evs, fr in zip(script, sizes):
                if evs:
                    ob.note_events(T.note_events(evs))
                want.append(ob.render(fr))
            want = np.concatenate(want, axis=1)
            assert np.sqrt(np.mean(want ** 2)) > 1e-2
            outs = {}
            for form in ("tp", "any", "split", "per-kind"):
                gpu_ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                gpu_ctx.split_max_waves = (1 << 20) if form == "split" else 0
                gpu_ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
                s = E.WelshSynth(gpu_ctx, params)
                blk = gpu_ctx.block(n, 256)
                got = []
                for evs, fr in zip(script, sizes):
                    if evs:
                        s.handle_midi_events(T.note_events(evs))
                    s.generate_batch_values(blk, fr)
                    got.append(blk.download(fr))
                got = np.concatenate(got, axis=1).astype(np.float64)
                rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
                assert int((rms > 1e-5).sum()) <= 1 and rms.max() <= 2e-4, (seed, form, int(np.argmax(rms)), float(rms.max()))
                outs[form] = got
                s.destroy(); blk.destroy()
            assert np.array_equal(outs["split"], outs["any"]), seed
            assert np.abs(outs["tp"] - outs["any"]).max() <= 2e-6 * max(1.0, np.abs(outs["any"]).max()), seed
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old
    assert gpu_ctx.debug_info()["zero_segments"] == 0
