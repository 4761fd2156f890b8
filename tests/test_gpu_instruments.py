"""GPU parity: FM voices (a6), sampler / drumkit voices (a7), mix bus (a15), WAV sink (a18)."""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T
from tests.seeds import drawn_seeds

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["serial", "time-parallel"])
def kernel_form(request, gpu_ctx):
    """FM and sampler banks have two forms as well (one voice per lane, kernels.h; one wavefront per voice with the
    block's frames over its lanes, welsh_tp.h): every test of this module runs against both."""
    old = gpu_ctx.time_parallel_max_voices
    gpu_ctx.time_parallel_max_voices = 0 if request.param == "serial" else old
    yield request.param
    gpu_ctx.time_parallel_max_voices = old


@pytest.mark.parametrize("form", ["as-set"])
def test_fm_per_voice_parity(gpu_ctx, oracle, form):
    """16 FM patches (beta 0.1 .. 15, through-zero FM), 60 blocks (ragged lengths), note-off at block 30, re-trigger at
    block 45; both forms of the FM render (one voice per lane, kernels.h; one wavefront per voice with the carrier
    phase as a prefix sum over its lanes, welsh_tp.h).  Tolerance: per-voice RMS <= 1e-5 vs the f64 oracle; the
    voice state of the two forms agrees bit for bit apart from the last bits of the carrier phase."""
    from groove_amd import entities as E
    _fm_parity(gpu_ctx, oracle, E)


def _fm_parity(gpu_ctx, oracle, E):
    n, frames, blocks = 50, 256, 60
    params = P.fm_voices(n)
    synth = E.FmSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, frames)
    ob = oracle.Bank.fm(params)
    on, off = P.note_on_all(n), P.note_off_all(n)
    got, want = [], []
    for b in range(blocks):
        if b in (0, 45):
            synth.handle_midi_events(on); ob.note_events(on)
        if b == 30:
            synth.handle_midi_events(off); ob.note_events(off)
        fr = [256, 256, 100, 7, 1, 255][b % 6]
        synth.generate_batch_values(block, fr)
        got.append(block.download(fr)); want.append(ob.render(fr))
    got = np.concatenate(got, axis=1).astype(np.float64); want = np.concatenate(want, axis=1)
    for v in range(n):
        rms = np.sqrt(np.mean((got[:, :, v] - want[:, :, v]) ** 2))
        assert np.sqrt(np.mean(want[:, :, v] ** 2)) > 1e-3
        assert rms <= 1e-5, f"fm voice {v}: rms {rms:.3e}"
    synth.destroy(); block.destroy()


def test_fm_index_of_the_reference_demo_projects(gpu_ctx, oracle, kernel_form):
    """The parameter sets of the reference's FM demo projects (projects/demos/instruments/fm-synthesizer-beta-*.json: ratio 2, depth 1, beta
    0 / 0.1 / 1 / 10 / 100, flat envelopes, one long A4) on the demo's key and others up to 93.  An index of 100 on A4 swings the carrier to
    101 x 440 Hz — more than a whole turn per frame; the oracle's phase drops whole turns (`pos -= floor(pos)`), and so must the device's
    64-bit counter (dsp_core.h turns_to_inc; until the end of round 5 it converted 2^64 and more to u64: every key from 69 up played 0.74
    off at beta 100).  Both forms of the FM render, 172 blocks, <= 1e-5 RMS per voice."""
    from groove_amd import entities as E
    betas, keys = (0.0, 0.1, 1.0, 10.0, 100.0), (45, 57, 60, 69, 72, 81, 93, 100)
    ps, ks = [], []
    for beta in betas:
        for key in keys:
            p = T.FmParams()
            p.ratio, p.depth, p.beta = 2.0, 1.0, beta
            p.carrier_envelope = T.EnvelopeParams(0.0, 0.0, 1.0, 0.0)
            p.modulator_envelope = T.EnvelopeParams(0.0, 0.0, 1.0, 0.0)
            p.dca_gain, p.dca_pan = 1.0, 0.0
            ps.append(p); ks.append(key)
    n = len(ps)
    params, keys_np, lanes = (T.FmParams * n)(*ps), np.array(ks, dtype=np.uint8), np.arange(n, dtype=np.uint32)
    synth, ob = E.FmSynth(gpu_ctx, params), oracle.Bank.fm(params)
    block = gpu_ctx.block(n, 256)
    ev = T.note_events_np(lanes, keys_np, True)
    synth.handle_midi_events(ev); ob.note_events(ev)
    got, want = [], []
    for b in range(172):
        synth.generate_batch_values(block, 256)
        got.append(block.download(256)); want.append(ob.render(256))
    got = np.concatenate(got, axis=1).astype(np.float64); want = np.concatenate(want, axis=1)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    assert np.sqrt(np.mean(want ** 2, axis=(0, 1))).min() > 0.1
    assert rms.max() <= 1e-5, (int(np.argmax(rms)), float(rms.max()), betas[int(np.argmax(rms)) // len(keys)], ks[int(np.argmax(rms))])
    synth.destroy(); block.destroy()


def test_fm_four_voices_per_wavefront_parity(gpu_ctx, oracle, kernel_form):
    """The FM kernel's four-voices-per-wavefront form (16 lanes x 16 frames per voice; banks of 4,096 voices and more by
    default) forced for a 50-voice bank by GROOVE_FM_TP_VPW4_MIN_VOICES=1 in a context of its own: the per-voice parity test
    above, and fused render + mix against the one-voice form's bus."""
    import os
    from groove_amd import entities as E
    if kernel_form == "serial":
        pytest.skip("a form of the time-parallel kernel")
    os.environ["GROOVE_FM_TP_VPW4_MIN_VOICES"] = "1"
    try:
        ctx2 = E.Context(0)
    finally:
        del os.environ["GROOVE_FM_TP_VPW4_MIN_VOICES"]
    try:
        probe = E.FmSynth(ctx2, P.fm_voices(50))
        assert "four voices per wavefront" in probe.kernel_form(256, True)
        probe.destroy()
        _fm_parity(ctx2, oracle, E)
        n = 203  # not a multiple of the 16 voices per workgroup
        params = P.fm_voices(n)
        on = P.note_on_all(n)
        buses = []
        for c in (ctx2, gpu_ctx):
            s = E.FmSynth(c, params)
            s.handle_midi_events(on)
            bus = c.bus(6 * 256)
            at = 0
            for fr in (256, 100, 256, 7, 256, 255):
                s.render_mix(bus, fr, at_frame=at)
                at += fr
            buses.append(bus.download()[:at].astype(np.float64))
            s.destroy(); bus.destroy()
        assert np.sqrt(np.mean(buses[1] ** 2)) > 1e-3
        assert np.abs(buses[0] - buses[1]).max() <= 2e-5 * max(1.0, float(np.abs(buses[1]).max()))
    finally:
        ctx2.close()


def test_fm_forms_leave_the_same_state(gpu_ctx, kernel_form):
    from groove_amd import entities as E
    if kernel_form == "serial":
        pytest.skip("compares the two forms itself")
    n = 37
    params = P.fm_voices(n)
    on, off = P.note_on_all(n), P.note_off_all(n)
    old = gpu_ctx.time_parallel_max_voices
    a, b = E.FmSynth(gpu_ctx, params), E.FmSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, 256)
    try:
        for blk in range(40):
            for s in (a, b):
                if blk in (0, 30): s.handle_midi_events(on)
                if blk == 12: s.handle_midi_events(off)
            fr = [256, 100, 7, 1][blk % 4]
            gpu_ctx.time_parallel_max_voices = old
            a.generate_batch_values(block, fr); xa = block.download(fr); sa = a.download_state()
            gpu_ctx.time_parallel_max_voices = 0
            b.generate_batch_values(block, fr); xb = block.download(fr); sb = b.download_state()
            assert np.max(np.abs(xa.astype(np.float64) - xb)) <= 2e-6, blk
            # FmState words: carrier {u64 phase, x1, x2}, modulator {...}, c_inc, m_inc (u64), 2 x EnvState (7 words), vflags, pad
            exact = list(range(2, 27))
            assert np.array_equal(sa[exact], sb[exact]), blk
            pa = sa[0].astype(np.uint64) | (sa[1].astype(np.uint64) << np.uint64(32))
            pb = sb[0].astype(np.uint64) | (sb[1].astype(np.uint64) << np.uint64(32))
            assert np.array_equal(pa, pb), blk   # same increments, integer sums: the carrier phase is exact too
    finally:
        gpu_ctx.time_parallel_max_voices = old
    a.destroy(); b.destroy(); block.destroy()


def test_sampler_is_exact_fetch(gpu_ctx, oracle):
    """Pointer stepping without interpolation is a pure fetch: the fp32 output must equal the
    oracle's value bit for bit (drumkit step 1 and pitched steps, staggered note-ons, one-shot
    end-of-buffer stop)."""
    from groove_amd import entities as E
    n, frames = 240, 256
    pcm, descs, lengths = P.drum_bank(scale=0.05)
    params = P.sampler_voices(n)
    s = E.Sampler(gpu_ctx, pcm, descs, params)
    ob = oracle.Bank.sampler(pcm, descs, params)
    block = gpu_ctx.block(n, frames)
    keys = P.sampler_keys(n)
    start = P.sampler_start_block(n) % 6
    for b in range(24):
        idx = np.nonzero(start == b)[0].astype(np.uint32)
        if idx.size:
            ev = T.note_events_np(idx, keys[idx], True)
            s.handle_midi_events(ev); ob.note_events(ev)
        s.generate_batch_values(block, frames)
        got = block.download(frames)
        want = ob.render(frames).astype(np.float32)
        assert np.array_equal(got, want), f"block {b}: sampler fetch differs"
    s.destroy(); block.destroy()


def test_sampler_ragged_blocks_materialised_and_fused(gpu_ctx, oracle):
    """Block lengths that are not multiples of the sampler's 16-frame fetch chunk or of the fused epilogue's
    8-frame flush (including 1 and 0): the materialised block stays bit-exact, the fused bus matches the
    oracle's sum; samples run off their ends inside chunks (scaled-down bank)."""
    from groove_amd import entities as E
    n = 240
    pcm, descs, lengths = P.drum_bank(scale=0.02)
    params = P.sampler_voices(n)
    keys = P.sampler_keys(n)
    on = T.note_events_np(np.arange(n, dtype=np.uint32), keys, True)
    a, f = E.Sampler(gpu_ctx, pcm, descs, params), E.Sampler(gpu_ctx, pcm, descs, params)
    oa, of = oracle.Bank.sampler(pcm, descs, params), oracle.Bank.sampler(pcm, descs, params)
    for x in (a, f, oa, of):
        (x.handle_midi_events if hasattr(x, "handle_midi_events") else x.note_events)(on)
    block = gpu_ctx.block(n, 256)
    bus = gpu_ctx.bus(256)
    for frames in (100, 7, 1, 16, 17, 0, 255, 33, 8, 256, 9):
        a.generate_batch_values(block, frames)
        want = oa.render(frames).astype(np.float32)
        if frames:
            assert np.array_equal(block.download(frames), want), frames
        f.render_mix(bus, frames)
        wb = of.render_bus(frames)
        if frames:
            assert np.max(np.abs(bus.download(frames).astype(np.float64) - wb)) <= 1e-4, frames
    for x in (a, f, block, bus):
        x.destroy()


def test_mix_bus_known_answers(gpu_ctx):
    """Restates gather_audio_basic / gather_audio / gather_audio_2 / _with_branches
    (orchestrator.rs:1444-1668) on constant-level blocks: sums of sources, Gain ceilings in
    chains, fan-in to an effect."""
    from groove_amd import entities as E
    frames = 64

    def const_block(levels):
        n = len(levels)
        b = gpu_ctx.block(n, frames)
        host = np.empty((2, frames, n), dtype=np.float32)
        host[:] = np.asarray(levels, dtype=np.float32)[None, None, :]
        b.upload(host)
        return b

    bus = gpu_ctx.bus(frames)
    # nothing patched → silence
    gpu_ctx.mix([], frames, bus)
    assert not bus.download().any()
    # single source, then two sources: 0.1, 0.2, 0.1 + 0.2
    b1, b2 = const_block([0.1]), const_block([0.2])
    gpu_ctx.mix([b1], frames, bus); assert np.allclose(bus.download(), 0.1, atol=1e-7)
    gpu_ctx.mix([b2], frames, bus); assert np.allclose(bus.download(), 0.2, atol=1e-7)
    gpu_ctx.mix([b1, b2], frames, bus); assert np.allclose(bus.download(), np.float32(0.1) + np.float32(0.2), atol=1e-7)
    # Gain{ceiling 0.5} on 0.1, siblings 0.2 0.3 0.4 → 0.1*0.5 + 0.2 + 0.3 + 0.4
    g = E.Effect(gpu_ctx, T.FX_GAIN, (T.FxParams * 1)(T.fx_params(ceiling=0.5)))
    c = const_block([0.1]); g.transform_audio(c, frames)
    sib = const_block([0.2, 0.3, 0.4])
    gpu_ctx.mix([c, sib], frames, bus)
    assert np.allclose(bus.download(), 0.1 * 0.5 + 0.2 + 0.3 + 0.4, atol=2e-7)
    # chains: 0.1*0.2*0.4, 0.3*0.6, 0.5*0.8 as three lanes, two gain stages
    lanes = const_block([0.1, 0.3, 0.5])
    g1 = E.Effect(gpu_ctx, T.FX_GAIN, (T.FxParams * 3)(T.fx_params(ceiling=0.2), T.fx_params(ceiling=0.6), T.fx_params(ceiling=0.8)))
    g2 = E.Effect(gpu_ctx, T.FX_GAIN, (T.FxParams * 3)(T.fx_params(ceiling=0.4), T.fx_params(ceiling=1.0), T.fx_params(ceiling=1.0)))
    g1.transform_audio(lanes, frames); g2.transform_audio(lanes, frames)
    out = lanes.download(frames)
    f32 = np.float32
    assert out[0, 0, 0] == f32(f32(0.1) * f32(0.2)) * f32(0.4)
    assert out[0, 0, 1] == f32(0.3) * f32(0.6) and out[0, 0, 2] == f32(0.5) * f32(0.8)
    # fan-in: 0.1 + 0.5 * (0.3 + 0.5): the effect sums its sources, then transforms once
    srcs = const_block([0.3, 0.5])
    tmp_bus = gpu_ctx.bus(frames)
    gpu_ctx.mix([srcs], frames, tmp_bus)               # sum of the effect's sources
    summed = gpu_ctx.block(1, frames)
    host = tmp_bus.download().T.reshape(2, frames, 1).copy()
    summed.upload(host)
    g.transform_audio(summed, frames)                  # Gain 0.5 applied once to the sum
    gpu_ctx.mix([const_block([0.1]), summed], frames, bus)
    assert np.allclose(bus.download(), 0.1 + 0.5 * (0.3 + 0.5), atol=2e-7)


def test_mix_large_rows_and_accumulate(gpu_ctx):
    """Row sums over many lanes (float4 and scalar paths, ragged n) against a float64 sum."""
    rng = np.random.default_rng(7)
    for n in (1, 3, 64, 1000, 16384 + 4, 100003):
        frames = 16
        host = rng.standard_normal((2, frames, n)).astype(np.float32)
        b = gpu_ctx.block(n, frames)
        b.upload(host)
        bus = gpu_ctx.bus(frames)
        gpu_ctx.mix([b], frames, bus)
        want = host.astype(np.float64).sum(axis=2).T
        got = bus.download().astype(np.float64)
        assert np.max(np.abs(got - want)) <= 2e-7 * np.sqrt(n) * 4 + 1e-6
        gpu_ctx.mix([b], frames, bus, accumulate=True)
        assert np.max(np.abs(bus.download() - 2 * want)) <= 1e-5 * max(1.0, np.abs(want).max())
        b.destroy(); bus.destroy()


def test_wav_sink_quantisation(gpu_ctx, oracle):
    """(x * 32767) as i16: truncation toward zero, saturation (helpers.rs:79-91)."""
    vals = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 1.5, -1.5, 3.05e-5, -3.05e-5, 0.99999, -0.99999, 2.9e-5,
                     np.nan, 1e9], dtype=np.float32)
    frames = vals.size // 2
    bus = gpu_ctx.bus(frames)
    import ctypes as C
    from groove_amd import lib
    lib.check(gpu_ctx.L.groove_upload(gpu_ctx.h, bus.ptr, vals.ctypes.data_as(C.POINTER(C.c_float)), vals.size), gpu_ctx.h)
    got = bus.to_i16(frames).reshape(-1)
    L = oracle.lib()
    want = np.array([L.oracle_wav_quantise(float(v)) for v in vals], dtype=np.int16)
    assert np.array_equal(got, want)
    assert list(got[:7]) == [0, 32767, -32767, 16383, -16383, 32767, -32768]


@pytest.mark.parametrize("n", [150, 3000, 20000])
def test_sampler_forms_leave_the_same_state(gpu_ctx, kernel_form, n):
    """Serial chunks of 16 fetches against the time-parallel gather: same samples, same voice state, bit for bit
    (staggered note-ons, pitched and drumkit buffers, voices running off their buffers, ragged block lengths).  The
    time-parallel form gives a wavefront ceil(n / 1024) adjacent voices: 1, 3 and 20 here (the last wave partly filled)."""
    from groove_amd import entities as E
    if kernel_form == "serial":
        pytest.skip("compares the two forms itself")
    pcm, descs, _ = P.drum_bank(scale=0.05)
    params = P.sampler_voices(n)
    keys = P.sampler_keys(n)
    old = gpu_ctx.time_parallel_max_voices
    a, b = E.Sampler(gpu_ctx, pcm, descs, params), E.Sampler(gpu_ctx, pcm, descs, params)
    block = gpu_ctx.block(n, 256)
    try:
        for blk in range(40):
            voices = np.arange(n, dtype=np.uint32)[(np.arange(n) % 13) == (blk % 13)]
            if blk < 26 and len(voices):
                ev = T.note_events_np(voices, keys[voices], True)
                a.handle_midi_events(ev); b.handle_midi_events(ev)
            fr = [256, 100, 7, 1, 255][blk % 5]
            gpu_ctx.time_parallel_max_voices = old
            a.generate_batch_values(block, fr); xa = block.download(fr); sa = a.download_state()
            gpu_ctx.time_parallel_max_voices = 0
            b.generate_batch_values(block, fr); xb = block.download(fr); sb = b.download_state()
            if not np.array_equal(xa, xb):
                bad = np.unique(np.nonzero(xa != xb)[2])
                raise AssertionError(f"block {blk}: {len(bad)} voices differ, first {bad[:8]}")
            assert np.array_equal(sa[:5], sb[:5]), blk   # idx, step (u64 each), playing
        assert np.abs(xa).max() >= 0.0
    finally:
        gpu_ctx.time_parallel_max_voices = old
    a.destroy(); b.destroy(); block.destroy()


def test_random_fm_patches_against_the_oracle(gpu_ctx, oracle, kernel_form):
    """FM patches DRAWN from a seed — ratio 0.25 - 9 (the benchmark's are all 2), depth 0 - 1, beta 0.05 - 20 (through-zero FM well past the
    benchmark's 15), envelopes with instant attacks and zero sustains — on random keys, ragged blocks, a note-off and a re-trigger: both
    forms of the FM render against the f64 oracle voice by voice, <= 1e-5 RMS up to a modulation index (beta x depth) of 5, rising with the
    index to 2e-5 at 10 and above (the bar tests/test_gpu_independent.py gives the carrier's fp32 sine at that depth): what the fp32
    modulator and its envelope leave is multiplied by the index on its way into the carrier's phase (seed 20165: key 89 x ratio 7.986 puts
    the modulator 1 % above SR/4, index 6.8, 1.14e-5 — gone one key up or down, or with the attack 1 % longer; docs/HISTORY.md section 10)."""
    import os
    from groove_amd import entities as E
    n, blocks = 48, 40
    lanes = np.arange(n, dtype=np.uint32)
    for seed in drawn_seeds(6):   # (400 seeds ran clean at the end of round 5)
        rng = np.random.default_rng(seed)
        ps = []
        for _ in range(n):
            p = T.FmParams()
            p.ratio, p.depth, p.beta = float(rng.choice([0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 7.0, rng.uniform(0.3, 9.0)])), float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.05, 20.0))
            env = lambda: T.EnvelopeParams(0.0 if rng.random() < 0.2 else float(rng.uniform(0.001, 0.3)), float(rng.uniform(0.05, 1.5)),   # noqa: E731
                                           0.0 if rng.random() < 0.15 else float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.05, 1.0)))
            p.carrier_envelope, p.modulator_envelope = env(), env()
            if p.carrier_envelope.sustain == 0.0 and p.carrier_envelope.decay < 0.3:
                p.carrier_envelope.decay = 0.3
            p.dca_gain, p.dca_pan = float(rng.uniform(0.3, 1.0)), float(rng.uniform(-1.0, 1.0))
            ps.append(p)
        params = (T.FmParams * n)(*ps)
        keys = rng.integers(30, 96, size=n).astype(np.uint8)
        synth, ob = E.FmSynth(gpu_ctx, params), oracle.Bank.fm(params)
        block = gpu_ctx.block(n, 256)
        got, want = [], []
        for b in range(blocks):
            if b in (0, 30):
                ev = T.note_events_np(lanes, keys, True)
                synth.handle_midi_events(ev); ob.note_events(ev)
            if b == 20:
                ev = T.note_events_np(lanes, keys, False)
                synth.handle_midi_events(ev); ob.note_events(ev)
            fr = int(rng.choice([256, 256, 256, 100, 7, 1]))
            synth.generate_batch_values(block, fr)
            got.append(block.download(fr)); want.append(ob.render(fr))
        got = np.concatenate(got, axis=1).astype(np.float64); want = np.concatenate(want, axis=1)
        rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
        bar = np.array([float(np.clip(2e-6 * p.beta * p.depth, 1e-5, 2e-5)) for p in ps])   # (1e-5 to an index of 5, 2e-5 from 10)
        assert np.isfinite(got).all() and (rms <= bar).all(), (seed, kernel_form, int(np.argmax(rms / bar)), float((rms / bar).max()))
        synth.destroy(); block.destroy()


def test_random_sampler_walks_are_exact(gpu_ctx, oracle, kernel_form):
    """Seeded random sampler walks: voices over the (scaled-down) synthetic bank with random sample, gain and one-shot flag, random note-ons
    (any key: pitched steps from 0.1 to 12 samples per frame) and note-offs on a few voices per block, ragged blocks: pointer stepping without
    interpolation is a fetch, so every block equals the oracle's bit for bit, in both kernel forms, samples running off their ends included."""
    import os
    from groove_amd import entities as E
    pcm, descs, lengths = P.drum_bank(scale=0.03)
    n = 200
    for seed in drawn_seeds(6):   # (300 seeds ran clean at the end of round 5)
        rng = np.random.default_rng(seed)
        params = (T.SamplerParams * n)(*[T.SamplerParams(int(rng.integers(len(descs))), int(rng.random() < 0.7), float(rng.uniform(0.1, 1.0))) for _ in range(n)])
        s, ob = E.Sampler(gpu_ctx, pcm, descs, params), oracle.Bank.sampler(pcm, descs, params)
        block = gpu_ctx.block(n, 256)
        for b in range(30):
            k = int(rng.integers(0, 24)) if b else n // 2
            if k:
                voices = np.sort(rng.choice(n, size=k, replace=False)).astype(np.uint32)
                ev = T.note_events_np(voices, rng.integers(24, 108, size=k).astype(np.uint8), bool(b == 0 or rng.random() < 0.75))
                s.handle_midi_events(ev); ob.note_events(ev)
            fr = int(rng.choice([256, 256, 256, 100, 33, 17, 7, 1]))
            s.generate_batch_values(block, fr)
            assert np.array_equal(block.download(fr), ob.render(fr).astype(np.float32)), (seed, kernel_form, b, fr)
        s.destroy(); block.destroy()
