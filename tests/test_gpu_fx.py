"""GPU parity for the effect kernels (a3, a4, a8-a12) vs the f64 oracle.

Inputs are deterministic pseudo-audio blocks; the same fp32 data feeds both sides.
Tolerances: Bitcrusher bit-exact; Gain exact (one fp32 multiply); IIR / delay-line effects
max abs error <= 2e-6 relative to a unit-scale input (fp32 rings, f64 recurrences).
"""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T
from tests.seeds import drawn_seeds

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


def _audio(n, frames_total, seed=1):
    rng = np.random.default_rng(seed)
    t = np.arange(frames_total)[None, :, None]
    f = (110.0 * 2.0 ** (np.arange(n) % 37 / 12.0))[None, None, :]
    x = 0.5 * np.sin(2 * np.pi * f * t / 44100.0 + np.arange(2)[:, None, None]) + 0.1 * rng.standard_normal((2, frames_total, n))
    x[:, frames_total // 2:, :] *= 0.0  # let tails ring out
    return x.astype(np.float32)


def _params(n, **kw):
    arr = (T.FxParams * n)()
    for i in range(n):
        arr[i] = T.fx_params(**{k: (v[i] if isinstance(v, (list, np.ndarray)) else v) for k, v in kw.items()})
    return arr


def _run(gpu_ctx, oracle, kind, params, x, frames=256, block_sizes=None):
    from groove_amd import entities as E
    n = x.shape[2]
    fx = E.Effect(gpu_ctx, kind, params)
    ofx = oracle.Fx(kind, params)
    block = gpu_ctx.block(n, frames)
    got, want = [], []
    pos = 0
    sizes = block_sizes or [frames] * (x.shape[1] // frames)
    for fr in sizes:
        chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
        block.upload(chunk)
        fx.transform_audio(block, fr)
        got.append(block.download(fr))
        want.append(ofx.process(chunk.astype(np.float64)))
        pos += fr
    fx.destroy(); block.destroy()
    return np.concatenate(got, axis=1), np.concatenate(want, axis=1)


def test_gain_exact(gpu_ctx, oracle):
    n = 100
    x = _audio(n, 512)
    ceil = np.linspace(0.0, 1.0, n).astype(np.float32)
    got, want = _run(gpu_ctx, oracle, T.FX_GAIN, _params(n, ceiling=list(ceil)), x)
    assert np.array_equal(got, (x * ceil[None, None, :]).astype(np.float32))
    assert np.max(np.abs(got - want)) <= 1e-7


@pytest.mark.parametrize("bits", [0, 1, 4, 8, 13, 15])
def test_bitcrusher_bit_exact(gpu_ctx, oracle, bits):
    """Integer quantise on the 16-bit scale: GPU == oracle for every sample, bit for bit."""
    n = 70
    x = _audio(n, 512, seed=bits)
    x[0, :8, 0] = [0.0, -0.0, 1.0, -1.0, 3.0e-5, -3.0e-5, 0.999985, 2.5]
    got, want = _run(gpu_ctx, oracle, T.FX_BITCRUSHER, _params(n, bits=bits), x)
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_biquad_lp12_and_hp12(gpu_ctx, oracle):
    n = 128
    x = _audio(n, 2048)
    cut = [40.0 + 150.0 * i for i in range(n)]
    for kind in (T.FX_BIQUAD_LP12, T.FX_BIQUAD_HP12):
        got, want = _run(gpu_ctx, oracle, kind, _params(n, cutoff_hz=cut, q=0.707), x)
        assert np.max(np.abs(got - want)) <= 2e-6


def test_biquad_other_modes(gpu_ctx, oracle):
    """Band-pass, band-stop, all-pass, peaking, low / high shelf (cookbook :113-198) — same kernel,
    host-side f64 coefficients."""
    n = 64
    x = _audio(n, 2048)
    cut = [200.0 + 120.0 * i for i in range(n)]
    cases = [(T.FX_BIQUAD_BP12, dict(bandwidth_hz=[30.0 + 10 * i for i in range(n)])),
             (T.FX_BIQUAD_BS12, dict(bandwidth_hz=[30.0 + 10 * i for i in range(n)])),
             (T.FX_BIQUAD_AP12, dict(q=[0.707 + 0.3 * i for i in range(n)])),
             (T.FX_BIQUAD_PEAK12, dict(db_gain=[-12.0 + 0.5 * i for i in range(n)])),
             (T.FX_BIQUAD_LSHELF12, dict(db_gain=[-12.0 + 0.5 * i for i in range(n)])),
             (T.FX_BIQUAD_HSHELF12, dict(db_gain=[-12.0 + 0.5 * i for i in range(n)]))]
    for kind, kw in cases:
        got, want = _run(gpu_ctx, oracle, kind, _params(n, cutoff_hz=cut, **kw), x)
        assert np.max(np.abs(want)) > 0.05
        assert np.max(np.abs(got - want)) <= 4e-6 * max(1.0, np.max(np.abs(want))), kind


def test_lp24_effect(gpu_ctx, oracle):
    n = 96
    x = _audio(n, 2048)
    cut = [40.0 * 500.0 ** (i / (n - 1)) for i in range(n)]
    rip = [0.707 + 0.03 * i for i in range(n)]
    got, want = _run(gpu_ctx, oracle, T.FX_BIQUAD_LP24, _params(n, cutoff_hz=cut, passband_ripple=rip), x)
    assert np.max(np.abs(got - want)) <= 2e-6


def test_delay_chorus_reverb(gpu_ctx, oracle):
    n = 72
    x = _audio(n, 256 * 40)
    got, want = _run(gpu_ctx, oracle, T.FX_DELAY, _params(n, delay_seconds=0.1), x)
    assert np.array_equal(got, want.astype(np.float32))  # a pure delay is exact
    assert np.array_equal(got[:, 4410:4410 + 256, :], x[:, :256, :])
    got, want = _run(gpu_ctx, oracle, T.FX_CHORUS, _params(n, voices=4, delay_seconds=0.25), x[:, :256 * 60 // 2, :])
    assert np.max(np.abs(got - want)) <= 2e-6
    got, want = _run(gpu_ctx, oracle, T.FX_REVERB, _params(n, attenuation=0.95, reverb_seconds=1.25), x)
    assert np.max(np.abs(want)) > 0.1
    assert np.max(np.abs(got - want)) <= 4e-6


def test_short_delay_and_ragged_blocks(gpu_ctx, oracle):
    """Delay shorter than a block (intra-block feedback path), odd block lengths, wet/dry mix."""
    n = 10
    x = _audio(n, 1500)
    sizes = [1, 63, 256, 100, 256, 256, 256, 256, 56]
    got, want = _run(gpu_ctx, oracle, T.FX_DELAY, _params(n, delay_seconds=0.001, wet=0.5), x, block_sizes=sizes)
    assert np.max(np.abs(got - want)) <= 1e-6
    got, want = _run(gpu_ctx, oracle, T.FX_REVERB, _params(n, attenuation=0.8, reverb_seconds=0.4, wet=0.3), x, block_sizes=sizes)
    assert np.max(np.abs(got - want)) <= 4e-6
    got, want = _run(gpu_ctx, oracle, T.FX_CHORUS, _params(n, voices=3, delay_seconds=0.002), x, block_sizes=sizes)
    assert np.max(np.abs(got - want)) <= 2e-6


def test_limiter_compressor_mixer(gpu_ctx, oracle):
    n = 33
    x = _audio(n, 512)
    got, want = _run(gpu_ctx, oracle, T.FX_LIMITER, _params(n, limit_min=0.1, limit_max=0.4), x)
    assert np.max(np.abs(got - want)) <= 1e-7
    got, want = _run(gpu_ctx, oracle, T.FX_COMPRESSOR, _params(n, limit_min=0.2, limit_max=0.25), x)
    assert np.max(np.abs(got - want)) <= 1e-7
    got, want = _run(gpu_ctx, oracle, T.FX_MIXER, _params(n), x)
    assert np.array_equal(got, x)


def test_effect_parameter_sets_of_the_reference_demo_projects(gpu_ctx, oracle):
    """The parameter sets the reference's own demo projects give its effects (projects/demos/effects/*.json; values only): 12 dB filters at
    1,000 Hz with q 0.707 and 20, bandwidths of 2, 30 and 2,000 Hz, shelf / peak gains of 6 and 30 dB, the swept low-pass's cutoffs from 20 kHz
    down to its clamp, the 24 dB low-pass at 1,000 Hz over the ripple sweep, gain ceilings, limiter windows, bitcrusher depths 8 and 13, the
    compressor's ratio 0.1 over its threshold ramp, chorus 4 x 0.25 s, delay 0.1 s, reverb 0.95 / 1.25 s — on a sine and on noise, the
    demos' two sources, against the oracle (IIR: 4e-6 of the larger of 1 and the output's peak; the integer and copy effects exact)."""
    frames_total = 8192
    rng = np.random.default_rng(5)
    t = np.arange(frames_total)
    sine = 0.8 * np.sin(2 * np.pi * 440.0 * t / 44100.0)
    noise = rng.uniform(-0.8, 0.8, frames_total)
    iir = [(T.FX_BIQUAD_LP12, [dict(cutoff_hz=1000.0, q=q) for q in (0.707, 20.0)] + [dict(cutoff_hz=c, q=1.0) for c in (20000.0, 5000.0, 300.0, 20.0, 0.0)]),
           (T.FX_BIQUAD_HP12, [dict(cutoff_hz=1000.0, q=q) for q in (0.707, 20.0)]),
           (T.FX_BIQUAD_AP12, [dict(cutoff_hz=1000.0, q=q) for q in (0.707, 20.0)]),
           (T.FX_BIQUAD_BP12, [dict(cutoff_hz=1000.0, bandwidth_hz=b) for b in (30.0, 2000.0)]),
           (T.FX_BIQUAD_BS12, [dict(cutoff_hz=1000.0, bandwidth_hz=b) for b in (2.0, 30.0, 2000.0)]),
           (T.FX_BIQUAD_PEAK12, [dict(cutoff_hz=1000.0, db_gain=g) for g in (6.0, 30.0)]),
           (T.FX_BIQUAD_LSHELF12, [dict(cutoff_hz=1000.0, db_gain=g) for g in (6.0, 30.0)]),
           (T.FX_BIQUAD_HSHELF12, [dict(cutoff_hz=1000.0, db_gain=g) for g in (6.0, 30.0)]),
           (T.FX_BIQUAD_LP24, [dict(cutoff_hz=1000.0, passband_ripple=r) for r in (0.1, 0.707, 1.332, 3.207, 6.332, 10.707)]),   # (the sweep: 0.707 + 10 v^2)
           (T.FX_REVERB, [dict(attenuation=0.95, reverb_seconds=1.25)]),
           (T.FX_CHORUS, [dict(voices=4, delay_seconds=0.25)]),
           (T.FX_COMPRESSOR, [dict(limit_min=th, limit_max=0.1) for th in (0.0, 0.25, 0.5, 1.0)])]
    exact = [(T.FX_GAIN, [dict(ceiling=c) for c in (0.1, 0.5, 1.0, 0.0)]),
             (T.FX_LIMITER, [dict(limit_min=a, limit_max=b) for a, b in ((0.1, 0.9), (0.4, 0.6))]),
             (T.FX_BITCRUSHER, [dict(bits=b) for b in (8, 13)]),
             (T.FX_DELAY, [dict(delay_seconds=0.1)])]
    for kind, sets in iir + exact:
        n = 2 * len(sets)   # each set on the sine and on the noise
        x = np.empty((2, frames_total, n), dtype=np.float32)
        x[:, :, 0::2] = sine[None, :, None]; x[:, :, 1::2] = noise[None, :, None]
        x[:, frames_total * 3 // 4:, :] = 0.0
        kws = [kw for kw in sets for _ in range(2)]
        params = (T.FxParams * n)(*[T.fx_params(**kw) for kw in kws])
        got, want = _run(gpu_ctx, oracle, kind, params, x)
        assert np.isfinite(got).all() and np.abs(want).max() > 0.05, kind
        if (kind, sets) in exact:
            assert np.array_equal(got, want.astype(np.float32)), (kind, float(np.abs(got - want).max()))
        else:
            err = np.abs(got - want).max(axis=(0, 1)); peak = np.maximum(1.0, np.abs(want).max(axis=(0, 1)))
            assert (err <= 4e-6 * peak).all(), (kind, kws[int(np.argmax(err / peak))], float((err / peak).max()))


def test_config3_chain_bus_parity(gpu_ctx, oracle):
    """Config #3 shape at a size the oracle finishes in seconds: 64 Welsh voices, each through
    BiQuad LP12 → Chorus → Delay → Reverb, summed on the bus; 40 blocks.  Bus/V RMS <= 1e-5."""
    from groove_amd import entities as E
    n, frames, blocks = 64, 256, 40
    params = P.welsh_voices(n)
    synth = E.WelshSynth(gpu_ctx, params)
    chain = P.chain_fx_params(n)
    fx = [E.Effect(gpu_ctx, k, p) for k, p in chain]
    ofx = [oracle.Fx(k, p) for k, p in chain]
    ob = oracle.Bank.welsh(params)
    on, off = P.note_on_all(n), P.note_off_all(n)
    block = gpu_ctx.block(n, frames)
    bus = gpu_ctx.bus(blocks * frames)
    want = []
    for b in range(blocks):
        if b == 0:
            synth.handle_midi_events(on); ob.note_events(on)
        if b == 20:
            synth.handle_midi_events(off); ob.note_events(off)
        synth.generate_batch_values(block, frames)
        for e in fx:
            e.transform_audio(block, frames)
        gpu_ctx.mix([block], frames, E._Slice(bus, b * frames))
        ref = ob.render(frames)
        for e in ofx:
            e.process(ref)
        want.append(oracle.mix(ref))
    got = bus.download().astype(np.float64) / n
    want = np.concatenate(want, axis=0) / n
    rms = np.sqrt(np.mean((got - want) ** 2))
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    assert rms <= 1e-5, f"chain bus rms {rms:.3e}"


def _second_context(env):
    import os
    from groove_amd import entities as E
    os.environ[env] = "1"
    try:
        return E.Context(0)
    finally:
        del os.environ[env]


@pytest.mark.parametrize("n", [1, 72, 1500])
def test_reverb_allpass_forms_ragged_blocks(gpu_ctx, oracle, n):
    """All-wet reverb.  The default form evaluates the two all-passes with no sequential step (each frame unrolls its
    feedback back to a ring slot of an earlier block; rings double-buffered); GROOVE_FX_CHUNKED_ALLPASS=1 selects the
    form that is time-parallel inside chunks of one line length (220 and 74 frames at 44.1 kHz), GROOVE_FX_SEQ_ALLPASS=1
    the sequential walk.  Block lengths around the line lengths and around multiples of them (blocks shorter than a line
    leave ring rows untouched: the copy rows of the direct form); lane counts with and without 16-byte accesses.  Same
    tolerance against the oracle, and the three forms bit-identical."""
    from groove_amd import entities as E
    sizes = [256, 1, 73, 74, 75, 148, 149, 219, 220, 221, 256, 255, 33, 256, 256]
    x = _audio(n, sum(sizes), seed=3)
    params = _params(n, attenuation=0.9, reverb_seconds=0.8)
    got, want = _run(gpu_ctx, oracle, T.FX_REVERB, params, x, block_sizes=sizes)
    assert np.max(np.abs(want)) > 0.1
    assert np.max(np.abs(got - want)) <= 4e-6
    for env in ("GROOVE_FX_SEQ_ALLPASS", "GROOVE_FX_CHUNKED_ALLPASS"):
        ctx2 = _second_context(env)
        fx = E.Effect(ctx2, T.FX_REVERB, params)
        block = ctx2.block(n, 256)
        other, pos = [], 0
        for fr in sizes:
            block.upload(np.ascontiguousarray(x[:, pos:pos + fr, :]))
            fx.transform_audio(block, fr)
            other.append(block.download(fr))
            pos += fr
        fx.destroy(); block.destroy(); ctx2.close()
        assert np.array_equal(got, np.concatenate(other, axis=1)), env


@pytest.mark.parametrize("n", [6, 64])
def test_chain_process_equals_stage_by_stage(gpu_ctx, n):
    """groove_fx_chain_process fuses the stages without feedback inside a block (gain, chorus, delay, reverb combs ...)
    into one pass; the result is the stage-by-stage result bit for bit, block after block, including partly wet stages,
    an IIR stage in the middle (which splits the run), a stage after the reverb, ragged blocks, and a reset."""
    from groove_amd import entities as E
    chain = [(T.FX_GAIN, _params(n, ceiling=[0.5 + 0.01 * i for i in range(n)])),
             (T.FX_CHORUS, _params(n, voices=2, delay_seconds=0.03, wet=[1.0 if i % 2 else 0.6 for i in range(n)])),
             (T.FX_BIQUAD_LP12, _params(n, cutoff_hz=[300.0 + 40 * i for i in range(n)], q=0.707)),
             (T.FX_DELAY, _params(n, delay_seconds=0.01, wet=0.8)),
             (T.FX_COMPRESSOR, _params(n, limit_min=0.2, limit_max=0.25)),
             (T.FX_DELAY, _params(n, delay_seconds=0.001, wet=0.7)),  # shorter than a block: the serial kernel, splits the run
             (T.FX_DELAY, _params(n, delay_seconds=0.02)),
             (T.FX_BITCRUSHER, _params(n, bits=12, wet=0.5)),
             (T.FX_CHORUS, _params(n, voices=2, delay_seconds=0.03)),
             (T.FX_REVERB, _params(n, attenuation=0.9, reverb_seconds=0.8)),
             (T.FX_LIMITER, _params(n, limit_min=0.0, limit_max=0.7))]
    sizes = [256, 256, 100, 256, 1, 255, 256, 256] * 3
    x = _audio(n, sum(sizes), seed=11)
    a = [E.Effect(gpu_ctx, k, p) for k, p in chain]
    b = [E.Effect(gpu_ctx, k, p) for k, p in chain]
    ba, bb = gpu_ctx.block(n, 256), gpu_ctx.block(n, 256)
    for rep in range(2):
        pos, peak = 0, 0.0
        for fr in sizes:
            chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
            ba.upload(chunk); bb.upload(chunk)
            for e in a:
                e.transform_audio(ba, fr)
            gpu_ctx.transform_chain(b, bb, fr)
            ga, gb = ba.download(fr), bb.download(fr)
            assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), (rep, pos)
            pos += fr
            peak = max(peak, float(np.max(np.abs(ga[:, :, 1::2]))))  # the fully wet lanes: only what went through every line
        assert peak > 1e-2
        for e in a + b:
            e.reset()
    with pytest.raises(RuntimeError, match="twice"):
        gpu_ctx.transform_chain([b[0], b[0]], bb, 256)
    for e in a + b:
        e.destroy()
    ba.destroy(); bb.destroy()


@pytest.mark.parametrize("n", [6, 64, 1000, 4100])
def test_chain_leaves_the_blocks_lane_sums_for_the_mix(gpu_ctx, n):
    """The last launch of a chain writes the block's lane sums (kernels.h fx_row_sum) and groove_mix reduces those instead
    of reading the block back: the bus must be the sum over the lanes of what the block holds — for chains that end in the
    fused run kernel, in the direct all-pass kernel, in a serial kernel (no sums: the mix reads the block), for ragged
    blocks, lane counts that are no multiple of 4 or of a workgroup, and after the caller wrote the block itself."""
    import ctypes as C
    from groove_amd import entities as E, lib
    chains = {
        "run": [(T.FX_GAIN, _params(n, ceiling=0.8)), (T.FX_DELAY, _params(n, delay_seconds=0.01, wet=0.7))],
        "reverb": [(T.FX_CHORUS, _params(n, voices=3, delay_seconds=0.03)), (T.FX_REVERB, _params(n, attenuation=0.9, reverb_seconds=0.5))],
        "reverb+limiter": [(T.FX_REVERB, _params(n, attenuation=0.9, reverb_seconds=0.5)), (T.FX_LIMITER, _params(n, limit_min=0.0, limit_max=0.6))],
        "serial-last": [(T.FX_GAIN, _params(n, ceiling=0.8)), (T.FX_BIQUAD_LP12, _params(n, cutoff_hz=900.0, q=0.707))],
        "mixer-last": [(T.FX_GAIN, _params(n, ceiling=0.8)), (T.FX_MIXER, _params(n))],
    }
    sizes = [256, 100, 256, 1, 255, 256]
    x = _audio(n, sum(sizes), seed=5)
    for name, chain in chains.items():
        fx = [E.Effect(gpu_ctx, k, p) for k, p in chain]
        blk, bus = gpu_ctx.block(n, 256), gpu_ctx.bus(256)
        pos = 0
        for fr in sizes:
            blk.upload(np.ascontiguousarray(x[:, pos:pos + fr, :]))
            gpu_ctx.transform_chain(fx, blk, fr)
            gpu_ctx.mix([blk], fr, bus)
            got = bus.download(fr).astype(np.float64)
            held = blk.download(fr).astype(np.float64)
            want = held.sum(axis=2).T  # [frames][2]
            scale = max(1.0, float(np.abs(held).sum(axis=2).max()))
            assert np.max(np.abs(got - want)) <= 2e-6 * scale, (name, n, fr, float(np.max(np.abs(got - want))))
            pos += fr
        # the caller writes the block through the raw pointer: the sums the chain left are stale, and the mix must not use them
        blk.upload(np.ascontiguousarray(x[:, :256, :]))
        gpu_ctx.transform_chain(fx, blk, 256)
        dev = blk.device_ptr()   # invalidates the sums (the pointer escapes)
        row = np.full(n, 0.25, dtype=np.float32)
        lib.check(gpu_ctx.L.groove_upload(gpu_ctx.h, C.c_void_p(dev), row.ctypes.data_as(C.POINTER(C.c_float)), n), gpu_ctx.h)  # frame 0, left
        gpu_ctx.mix([blk], 256, bus)
        assert abs(float(bus.download(1)[0, 0]) - 0.25 * n) <= 1e-4 * n, name
        # ... and when it keeps the pointer and writes again later, groove_block_mark_dirty says so
        gpu_ctx.transform_chain(fx, blk, 256)
        row[:] = -0.5
        lib.check(gpu_ctx.L.groove_upload(gpu_ctx.h, C.c_void_p(dev), row.ctypes.data_as(C.POINTER(C.c_float)), n), gpu_ctx.h)
        blk.mark_dirty()
        gpu_ctx.mix([blk], 256, bus)
        assert abs(float(bus.download(1)[0, 0]) + 0.5 * n) <= 1e-4 * n, name
        for e in fx:
            e.destroy()
        blk.destroy(); bus.destroy()


def test_rendered_block_written_through_its_pointer_is_mixed_as_written(gpu_ctx):
    """a render leaves the block's lane sums; a caller that scales the block through groove_block_device_ptr and
    then mixes it must hear the scaled block."""
    import ctypes as C
    from groove_amd import entities as E, lib, patches as P
    n = 512
    synth = E.WelshSynth(gpu_ctx, P.welsh_voices(n))
    synth.handle_midi_events(P.note_on_all(n))
    blk, bus = gpu_ctx.block(n, 256), gpu_ctx.bus(256)
    synth.generate_batch_values(blk, 256)
    gpu_ctx.mix([blk], 256, bus)
    before = bus.download().astype(np.float64)
    host = blk.download(256)
    dev = blk.device_ptr()
    scaled = np.ascontiguousarray(host * np.float32(0.5))
    lib.check(gpu_ctx.L.groove_upload(gpu_ctx.h, C.c_void_p(dev), scaled.ctypes.data_as(C.POINTER(C.c_float)), scaled.size), gpu_ctx.h)
    gpu_ctx.mix([blk], 256, bus)
    after = bus.download().astype(np.float64)
    assert np.abs(before).max() > 1e-2
    assert np.max(np.abs(after - 0.5 * before)) <= 1e-5 * max(1.0, np.abs(before).max())
    synth.destroy(); blk.destroy(); bus.destroy()


def test_reverb_long_blocks_and_sample_rate_change(oracle):
    """The direct all-pass form unrolls at most 8 hops per frame; a block longer than 8 x the shorter line (592 frames at
    44.1 kHz) takes the chunk-parallel form with the comb sum written in place — same result.  And a sample-rate change
    rebuilds the delay-line geometry (both copies of the all-pass rings): the effect then matches a fresh oracle effect
    at the new rate."""
    from groove_amd import entities as E
    n = 12
    ctx = E.Context(0)
    try:
        params = _params(n, attenuation=0.9, reverb_seconds=0.8)
        sizes = [1024, 256, 700, 1024, 100]
        x = _audio(n, sum(sizes), seed=5)
        fx, ofx = E.Effect(ctx, T.FX_REVERB, params), oracle.Fx(T.FX_REVERB, params)
        block = ctx.block(n, 1024)
        pos = 0
        for fr in sizes:
            chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
            block.upload(chunk)
            fx.transform_audio(block, fr)
            want = ofx.process(chunk.astype(np.float64))
            assert np.max(np.abs(block.download(fr) - want)) <= 4e-6, (pos, fr)
            pos += fr
        ctx.update_sample_rate(48000)
        ofx = oracle.Fx(T.FX_REVERB, params, sr=48000)
        pos, peak = 0, 0.0
        for fr in [256, 256, 300, 256, 256, 256, 256, 256, 256]:
            chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
            block.upload(chunk)
            fx.transform_audio(block, fr)
            want = ofx.process(chunk.astype(np.float64))
            assert np.max(np.abs(block.download(fr) - want)) <= 4e-6, ("48k", pos)
            peak = max(peak, float(np.max(np.abs(want))))
            pos += fr
        assert peak > 0.05  # (the shortest comb is 1,426 frames at 48 kHz: the tail starts in the sixth block)
        fx.destroy(); block.destroy()
    finally:
        ctx.close()


def test_random_chains_fused_equal_stage_by_stage_and_follow_the_oracle(gpu_ctx, oracle):
    """Seeded random effect chains — two to seven stages drawn from all seventeen kinds with random (uniform-geometry) parameters, 3 to 1,500
    lanes, ragged blocks, a reset in the middle, new parameters for a stage between blocks (automation): groove_fx_chain_process (stages grouped into fused runs, IIR stages in between, the reverb's
    all-passes behind its run) is the stage-by-stage result BIT FOR BIT, whatever the grouping; and chains of linear stages follow the f64
    oracle chain to 4e-6 of the signal's scale per stage."""
    import os
    from groove_amd import entities as E
    linear = [T.FX_GAIN, T.FX_BIQUAD_LP12, T.FX_BIQUAD_LP24, T.FX_CHORUS, T.FX_DELAY, T.FX_REVERB, T.FX_MIXER, T.FX_BIQUAD_HP12, T.FX_BIQUAD_BP12,
              T.FX_BIQUAD_BS12, T.FX_BIQUAD_AP12, T.FX_BIQUAD_PEAK12, T.FX_BIQUAD_LSHELF12, T.FX_BIQUAD_HSHELF12]
    nonlinear = [T.FX_BITCRUSHER, T.FX_LIMITER, T.FX_COMPRESSOR]

    def draw(rng, n, kinds):
        k = int(rng.choice(kinds))
        lane = lambda lo, hi: [float(v) for v in rng.uniform(lo, hi, size=n)]   # noqa: E731
        kw = dict(ceiling=lane(0.3, 1.0), cutoff_hz=lane(150.0, 6000.0), q=lane(0.5, 3.0), passband_ripple=lane(0.71, 3.0), wet=lane(0.3, 1.0),
                  attenuation=lane(0.5, 0.95), bandwidth_hz=lane(100.0, 2000.0), db_gain=lane(-9.0, 9.0), limit_min=0.05, limit_max=0.6,
                  bits=int(rng.integers(4, 15)), voices=int(rng.integers(1, 5)),
                  delay_seconds=float(rng.choice([0.0005, 0.002, 0.004, 0.012, 0.03])), reverb_seconds=float(rng.uniform(0.3, 1.5)))
        if k == T.FX_REVERB:
            kw["wet"] = 1.0 if rng.random() < 0.7 else lane(0.3, 1.0)      # (all wet: the combs ride in the fused run)
        return k, _params(n, **kw)

    seeds = drawn_seeds(8)   # (a campaign of 300 seeds ran clean at the end of round 5)
    for seed in seeds:
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.choice([3, 64, 130, 1000, 1500]))
        only_linear = rng.random() < 0.5
        chain = [draw(rng, n, linear if only_linear else linear + nonlinear) for _ in range(int(rng.integers(2, 8)))]
        sizes = [int(rng.choice([256, 256, 256, 100, 37, 255, 1])) for _ in range(14)]
        x = _audio(n, sum(sizes), seed=seed)
        a = [E.Effect(gpu_ctx, k, p) for k, p in chain]
        b = [E.Effect(gpu_ctx, k, p) for k, p in chain]
        o = [oracle.Fx(k, p) for k, p in chain] if only_linear else None
        ba, bb = gpu_ctx.block(n, 256), gpu_ctx.block(n, 256)
        pos, worst, scale = 0, 0.0, 1.0
        reset_at = int(rng.integers(4, 12))
        for i, fr in enumerate(sizes):
            if i == reset_at:
                for e in a + b:
                    e.reset()
                o = [oracle.Fx(k, p) for k, p in chain] if only_linear else None
            if i != reset_at and rng.random() < 0.25:   # automation between blocks: new parameters for one stage (same line geometry), all three chains
                j = int(rng.integers(len(chain)))
                k, old_p = chain[j]
                _, new_p = draw(rng, n, [k])
                for lane in range(n):                  # delay-line geometry cannot change after creation
                    new_p[lane].voices, new_p[lane].delay_seconds, new_p[lane].reverb_seconds = old_p[lane].voices, old_p[lane].delay_seconds, old_p[lane].reverb_seconds
                    if k == T.FX_REVERB:
                        new_p[lane].wet = old_p[lane].wet   # (all-wet or not decides which kernels a reverb takes: kept)
                chain[j] = (k, new_p)
                a[j].set_params(new_p); b[j].set_params(new_p)
                if only_linear:
                    o[j].set_params(new_p)
            chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
            ba.upload(chunk); bb.upload(chunk)
            for e in a:
                e.transform_audio(ba, fr)
            gpu_ctx.transform_chain(b, bb, fr)
            ga, gb = ba.download(fr), bb.download(fr)
            assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), (seed, i, [k for k, _ in chain])
            if only_linear:
                w = chunk.astype(np.float64)
                for e in o:
                    e.process(w)
                scale = max(scale, float(np.abs(w).max()))
                worst = max(worst, float(np.abs(gb.astype(np.float64) - w).max()))
            pos += fr
        if only_linear:
            assert worst <= 4e-6 * scale * len(chain), (seed, worst, scale, [k for k, _ in chain])
        for e in a + b:
            e.destroy()
        ba.destroy(); bb.destroy()


def test_random_linear_chains_at_other_sample_rates(oracle):
    """Effect chains drawn from a seed in a context of its own at 22,050 / 48,000 / 96,000 Hz: delay-line geometry (frames = seconds x rate),
    the reverb's comb and all-pass lengths and the RBJ / 24 dB coefficients are re-derived for the rate; the fused chain against the f64
    oracle chain at the same rate, 4e-6 of the signal's scale per stage."""
    import os
    from groove_amd import entities as E
    linear = [T.FX_GAIN, T.FX_BIQUAD_LP12, T.FX_BIQUAD_LP24, T.FX_CHORUS, T.FX_DELAY, T.FX_REVERB, T.FX_BIQUAD_HP12, T.FX_BIQUAD_BP12,
              T.FX_BIQUAD_PEAK12, T.FX_BIQUAD_LSHELF12, T.FX_BIQUAD_HSHELF12]
    for seed in drawn_seeds(3):   # (60 seeds ran clean at the end of round 5)
        rng = np.random.default_rng(7000 + seed)
        sr = int(rng.choice([22050, 48000, 96000]))
        n = int(rng.choice([5, 64, 600]))
        chain = []
        for _ in range(int(rng.integers(2, 6))):
            k = int(rng.choice(linear))
            lane = lambda lo, hi: [float(v) for v in rng.uniform(lo, hi, size=n)]   # noqa: E731
            chain.append((k, _params(n, ceiling=lane(0.3, 1.0), cutoff_hz=lane(150.0, 0.4 * sr), q=lane(0.5, 3.0), passband_ripple=lane(0.71, 3.0),
                                     wet=1.0 if k == T.FX_REVERB else lane(0.3, 1.0), attenuation=lane(0.5, 0.95), bandwidth_hz=lane(100.0, 2000.0),
                                     db_gain=lane(-9.0, 9.0), voices=int(rng.integers(1, 5)), delay_seconds=float(rng.choice([0.002, 0.004, 0.012, 0.03])),
                                     reverb_seconds=float(rng.uniform(0.3, 1.5)))))
        sizes = [int(rng.choice([256, 256, 256, 100, 37, 1])) for _ in range(12)]
        x = _audio(n, sum(sizes), seed=seed)
        ctx = E.Context(0)
        try:
            ctx.update_sample_rate(sr)
            fx = [E.Effect(ctx, k, p) for k, p in chain]
            o = [oracle.Fx(k, p, sr) for k, p in chain]
            blk = ctx.block(n, 256)
            pos, worst, scale = 0, 0.0, 1.0
            for fr in sizes:
                chunk = np.ascontiguousarray(x[:, pos:pos + fr, :])
                blk.upload(chunk)
                ctx.transform_chain(fx, blk, fr)
                w = chunk.astype(np.float64)
                for e in o:
                    e.process(w)
                scale = max(scale, float(np.abs(w).max()))
                worst = max(worst, float(np.abs(blk.download(fr).astype(np.float64) - w).max()))
                pos += fr
            assert worst <= 4e-6 * scale * len(chain), (seed, sr, worst, scale, [k for k, _ in chain])
            for e in fx:
                e.destroy()
            blk.destroy()
        finally:
            ctx.close()
