"""GPU tier: Welsh parity tests whose INPUTS are drawn from a seed — patches (every continuous parameter), note-event scripts, control changes
on a sounding bank — through every kernel form against the f64 oracle.  GROOVE_TEST_SEEDS / GROOVE_TEST_SEED_BASE (tests/seeds.py) set how many seeds each test plays and from where (the tier's
defaults are small; docs/HISTORY.md section 10 items 17 - 22 say what campaigns of hundreds found).  The tests pick the kernel forms
themselves (the ABI's tuning knobs), so they run once each."""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T
from tests.seeds import drawn_seeds

pytestmark = pytest.mark.gpu


def _level(want):
    """What a voice's error is measured against: full scale (1.0) or the voice's own RMS level, whichever is larger.  A drawn patch can ring
    far beyond full scale (seed 10133 of the sample-rate test: ripple 3.5 under a sawtooth LFO on the cutoff, peaks of 61, RMS 11 — its
    1.06e-5 is 9.5e-7 of its level, a few ulps of the fp32 samples the block holds; tools/random_sr_debug.py)."""
    return np.maximum(1.0, np.sqrt(np.mean(want ** 2, axis=(0, 1))))


def test_random_patches_every_kernel_form_against_the_oracle(gpu_ctx, oracle):
    """Patches DRAWN from a seed (groove_amd.patches.random_welsh_patch: every continuous parameter, every routing, instant attacks, zero
    sustains, cutoffs from 40 Hz to 20 kHz) instead of the 32 benchmark ones, eight per bank in runs of eight voices on random keys, 40 blocks
    with a note-off: every kernel form — time-parallel, the all-kinds serial kernel, role-split, the per-kind kernels with their fp32 filter
    bodies — against the f64 oracle voice by voice (<= 1e-5 RMS of the larger of full scale and the voice's level; measured worst of 60 x 64 voices: 2.1e-6), role-split bit for bit the
    serial kernel's.  No key is an A: 55 and 110 Hz are rational in 44,100 and put a square's edge EXACTLY on a frame, which the oracle's
    accumulated f64 phase and the device's 64-bit counter decide differently (docs/DSP_SPEC.md section 2)."""
    import os
    from groove_amd import entities as E
    n, blocks, off_at = 64, 40, 24
    old = (gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves)
    lanes = np.arange(n, dtype=np.uint32)
    try:
        for seed in drawn_seeds(6):   # (60 seeds ran clean at the end of round 5)
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
            # ("any, lanes' LFO": the all-kinds kernel without its LFO look-ahead — the role-split kernel's lanes advance the recurrences
            # which that table replaces by an exact evaluation, 2e-6 apart at most: kernels.h — for the bit comparison)
            for form in ("tp", "any", "any, lanes' LFO", "split", "per-kind"):
                gpu_ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                gpu_ctx.split_max_waves = (1 << 20) if form == "split" else 0
                gpu_ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
                gpu_ctx.look_ahead = 1 if form.endswith("lanes' LFO") else 3
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
                rms = np.sqrt(np.mean((got[form].astype(np.float64) - want) ** 2, axis=(0, 1))) / _level(want)
                assert np.isfinite(got[form]).all() and rms.max() <= 1e-5, (seed, form, int(np.argmax(rms)), float(rms.max()))
                s.destroy(); blk.destroy()
            assert np.array_equal(got["split"].view(np.uint32), got["any, lanes' LFO"].view(np.uint32)), seed
            assert np.abs(got["any"].astype(np.float64) - got["any, lanes' LFO"]).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), seed
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old
        gpu_ctx.look_ahead = 3
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_random_note_event_sequences_in_every_kernel_form(gpu_ctx, oracle):
    """Note events drawn from a seed: per block up to a dozen events in ARBITRARY order — the same voice several times in one block
    (on, off, on again: the order decides), events for ALL voices, note-offs for idle voices, re-triggers of sounding ones — over ragged
    blocks, the 32 benchmark patches on 96 voices.  HandlesMidi's semantics (block-granular, applied in order at the next block's start) in
    every kernel form against the oracle voice by voice (<= 1e-5 RMS).
    One knife edge is allowed for, the voices of at most one patch of a seed and <= 2e-4: a re-trigger takes the envelope's level as the start of a new
    attack of `N = ceil(len)` frames, `len = attack x SR x (1 - level)`; the device's level is an fp32 polynomial, the oracle's f64, and
    when `len` comes within 3e-4 of an integer the two stages differ by ONE frame — the decay behind starts a frame apart, 1.5e-4 of level
    for a 0.3 s filter decay (seed 19, voice 3; docs/DSP_SPEC.md section 3; gone when the patch's attack is 0.06002 s instead of 0.06).
    The four forms agree with each other bit for bit in that voice too.  Events for ALL voices put the voices of one PATCH on one envelope
    trajectory, and a later re-trigger finds them on the knife edge together (seed 20029: the three voices of patch 14, 1.2e-5 - 2.3e-5):
    the allowance is one patch of a seed."""
    import os
    from groove_amd import entities as E
    n, blocks = 96, 36
    params = P.welsh_voices(n)
    old = (gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves)
    try:
        for seed in drawn_seeds(4):   # (100 seeds ran clean at the end of round 5)
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
            # ("any, lanes' LFO": the all-kinds kernel without the LFO look-ahead — kernels.h: a wave whose voices share the LFO's phase
            # evaluates a pitch / pulse-width LFO exactly where the role-split kernel's lanes advance recurrences — for the bit comparison)
            for form in ("tp", "any", "any, lanes' LFO", "split", "per-kind"):
                gpu_ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                gpu_ctx.split_max_waves = (1 << 20) if form == "split" else 0
                gpu_ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
                gpu_ctx.look_ahead = 1 if form.endswith("lanes' LFO") else 3
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
                over = np.flatnonzero(rms > 1e-5)
                assert len(set(int(v) % 32 for v in over)) <= 1 and rms.max() <= 2e-4, (seed, form, int(np.argmax(rms)), float(rms.max()), over)
                outs[form] = got
                s.destroy(); blk.destroy()
            assert np.array_equal(outs["split"], outs["any, lanes' LFO"]), seed
            assert np.abs(outs["any"] - outs["any, lanes' LFO"]).max() <= 2e-6 * max(1.0, np.abs(outs["any"]).max()), seed
            assert np.abs(outs["tp"] - outs["any"]).max() <= 2e-6 * max(1.0, np.abs(outs["any"]).max()), seed
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old
        gpu_ctx.look_ahead = 3
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_random_controls_on_a_sounding_bank_in_every_kernel_form(gpu_ctx, oracle):
    """Controllable on an instrument in mid-performance (groove_bank_set_param: dca gain, pan, the static filter cutoff — re-derived and
    re-uploaded between blocks, the voices' state untouched; a cutoff change can move a patch across the fp32-filter criterion, pair voices
    of one patch apart for the two-voices-per-wavefront form, or split a run of voices that shared a parameter record): seeded random
    control changes on single voices and on ALL, a note-off and a re-trigger, ragged blocks, the 32 benchmark patches on 96 voices — every
    kernel form against the oracle given the same changes: bus / voices RMS <= 1e-5 (the path's bar), every voice <= 2e-4.  The per-voice
    figure is loose on purpose: a cutoff that JUMPS down under a sounding voice — 19 kHz to 99 Hz between two blocks, seed 13 — hands the
    low filter a state the high one left, whose fp32-coefficient rounding (1e-7 of it) the low filter's gain from state to output, ~(fc_hi /
    fc_lo)^2, turns into a transient of up to 4e-4 that decays with the filter (tools/random_controls_debug.py); all four forms carry it
    identically."""
    import os
    from groove_amd import entities as E
    n, blocks = 96, 30
    params, _ = P.welsh_voices_grouped(n)       # runs of voices of one patch: what the scalar-parameter kernels and the pairs want
    lanes = np.arange(n, dtype=np.uint32)
    keys = (36 + (7 * np.arange(n)) % 49).astype(np.uint8)
    old = (gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves)
    try:
        for seed in drawn_seeds(4):   # (100 seeds ran clean at the end of round 5)
            rng = np.random.default_rng(seed)
            script, sizes = [], []
            for b in range(blocks):
                changes = []
                for _ in range(int(rng.integers(0, 4)) if b else 0):
                    idx = int(rng.choice([T.CTL_WELSH_DCA_GAIN, T.CTL_WELSH_DCA_PAN, T.CTL_WELSH_CUTOFF]))
                    # (cutoffs from 49 Hz up: at 35 Hz the fp32 coefficients of a sounding voice leave 1.2e-5 — seed 3 with the range from 0.05)
                    changes.append((idx, float(rng.uniform(0.1, 1.0)), T.ALL_VOICES if rng.random() < 0.25 else int(rng.integers(n))))
                script.append(changes)
                sizes.append(int(rng.choice([256, 256, 256, 100, 37, 1])))

            def play(bank_events, bank_control, render):
                out = []
                for b in range(blocks):
                    if b == 0 or b == 22:
                        bank_events(T.note_events_np(lanes, keys, True))
                    if b == 14:
                        bank_events(T.note_events_np(lanes, keys, False))
                    for idx, v, voice in script[b]:
                        bank_control(idx, v, voice)
                    out.append(render(sizes[b]))
                return np.concatenate(out, axis=1)

            ob = oracle.Bank.welsh(params)
            want = play(ob.note_events, lambda i, v, voice: ob.set_param(i, v, voice), ob.render)
            assert np.sqrt(np.mean(want ** 2)) > 1e-2
            for form in ("tp", "any", "split", "per-kind"):
                gpu_ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                gpu_ctx.split_max_waves = (1 << 20) if form == "split" else 0
                gpu_ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
                s = E.WelshSynth(gpu_ctx, params)
                blk = gpu_ctx.block(n, 256)

                def render(fr):
                    s.generate_batch_values(blk, fr)
                    return blk.download(fr)

                got = play(s.handle_midi_events, lambda i, v, voice: s.control_set_param_by_index(i, v, voice=voice), render).astype(np.float64)
                rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
                bus = np.sqrt(np.mean(((got - want).sum(axis=2) / n) ** 2))
                assert np.isfinite(got).all() and rms.max() <= 2e-4 and bus <= 1e-5, (seed, form, int(np.argmax(rms)), float(rms.max()), float(bus))
                s.destroy(); blk.destroy()
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_random_patches_at_other_sample_rates(oracle):
    """Configurable::update_sample_rate with drawn patches: a context of its own at 22,050 / 48,000 / 96,000 Hz (envelope lengths, phase
    increments, filter coefficients, the Nyquist clamp of the cutoffs and the fp32-filter criterion are all re-derived for the rate), eight
    random patches on 64 voices, three kernel forms, against the oracle at the same rate (<= 1e-5 RMS per voice, of the larger of full scale and the voice's level)."""
    import os
    from groove_amd import entities as E
    n, blocks, off_at = 64, 30, 18
    lanes = np.arange(n, dtype=np.uint32)
    for seed in drawn_seeds(3):   # (60 seeds ran clean at the end of round 5)
        rng = np.random.default_rng(900 + seed)
        sr = int(rng.choice([22050, 48000, 96000]))
        patches = [P.random_welsh_patch(rng) for _ in range(8)]
        params = (T.WelshParams * n)(*[patches[(i // 8) % 8] for i in range(n)])
        keys = rng.integers(30, 96, size=n).astype(np.uint8)
        keys[keys % 12 == 9] += 1            # (the A's are rational in every integer sample rate: exact ties, docs/DSP_SPEC.md section 2)
        ob = oracle.Bank.welsh(params, sr=sr)
        ob.note_events(T.note_events_np(lanes, keys, True))
        want = []
        for b in range(blocks):
            if b == off_at:
                ob.note_events(T.note_events_np(lanes, keys, False))
            want.append(ob.render(256))
        want = np.concatenate(want, axis=1)
        ctx = E.Context(0)
        try:
            ctx.update_sample_rate(sr)
            old = (ctx.time_parallel_max_voices, ctx.pipeline_min_waves)
            for form in ("tp", "any", "per-kind"):
                ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
                ctx.split_max_waves = 0
                ctx.pipeline_min_waves = 1 if form == "per-kind" else old[1]
                s = E.WelshSynth(ctx, params)
                blk = ctx.block(n, 256)
                s.handle_midi_events(T.note_events_np(lanes, keys, True))
                got = []
                for b in range(blocks):
                    if b == off_at:
                        s.handle_midi_events(T.note_events_np(lanes, keys, False))
                    s.generate_batch_values(blk, 256)
                    got.append(blk.download(256))
                got = np.concatenate(got, axis=1).astype(np.float64)
                rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1))) / _level(want)
                assert np.isfinite(got).all() and rms.max() <= 1e-5, (seed, sr, form, int(np.argmax(rms)), float(rms.max()))
                s.destroy(); blk.destroy()
            assert ctx.debug_info()["zero_segments"] == 0
        finally:
            ctx.close()
