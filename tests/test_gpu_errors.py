"""GPU tier: error behaviour of the C ABI.  The reference's audio-path trait methods are infallible and
its graph edits return anyhow::Result (orchestrator.rs:263-304); the library returns a non-zero status
plus groove_last_error() and never aborts or falls back."""
import ctypes as C

import numpy as np
import pytest

from groove_amd import abi_types as T, lib, patches as P

pytestmark = pytest.mark.gpu


def test_argument_validation(gpu_ctx):
    from groove_amd import entities as E
    L = gpu_ctx.L
    params = P.welsh_voices(8)
    s = E.WelshSynth(gpu_ctx, params)
    small, big = gpu_ctx.block(4, 256), gpu_ctx.block(8, 64)
    with pytest.raises(lib.GrooveError, match="lanes"):
        s.generate_batch_values(small, 16)
    with pytest.raises(lib.GrooveError, match="capacity"):
        s.generate_batch_values(big, 65)
    with pytest.raises(lib.GrooveError, match="out of range"):
        s.handle_midi_events(T.note_events([(8, 60, True)]))
    with pytest.raises(lib.GrooveError, match="unknown control"):
        s.control_set_param_by_index(999, 0.5)
    with pytest.raises(lib.GrooveError, match="voice out of range"):
        s.control_set_param_by_index(T.CTL_WELSH_DCA_GAIN, 0.5, voice=99)
    # zero voices / zero-size blocks are errors, not crashes
    h = C.c_void_p()
    assert L.groove_welsh_create(gpu_ctx.h, params, 0, C.byref(h)) != 0
    assert L.groove_block_create(gpu_ctx.h, 0, 256, C.byref(h)) != 0
    assert L.groove_fx_create(gpu_ctx.h, 99, (T.FxParams * 1)(T.fx_params()), 1, C.byref(h)) != 0
    assert b"unknown effect kind" in L.groove_last_error(gpu_ctx.h)
    # rendering zero frames is a no-op
    s.generate_batch_values(big, 0)
    # delay-line geometry must be uniform across the lanes of one effect bank, and cannot change later
    two = (T.FxParams * 2)(T.fx_params(delay_seconds=0.1), T.fx_params(delay_seconds=0.2))
    with pytest.raises(lib.GrooveError, match="uniform"):
        E.Effect(gpu_ctx, T.FX_DELAY, two)
    d = E.Effect(gpu_ctx, T.FX_DELAY, (T.FxParams * 2)(T.fx_params(delay_seconds=0.1), T.fx_params(delay_seconds=0.1)))
    with pytest.raises(lib.GrooveError, match="geometry cannot change"):
        d.set_params((T.FxParams * 2)(T.fx_params(delay_seconds=0.3), T.fx_params(delay_seconds=0.3)))
    with pytest.raises(lib.GrooveError, match="lanes"):
        d.transform_audio(small, 16)
    # sampler descriptors are checked against the bank
    pcm = np.zeros(100, dtype=np.float32)
    descs = (T.SampleDesc * 1)()
    descs[0].offset, descs[0].length = 50, 100
    sp = (T.SamplerParams * 1)()
    with pytest.raises(lib.GrooveError, match="exceeds bank"):
        E.Sampler(gpu_ctx, pcm, descs, sp)
    descs[0].offset, descs[0].length = 0, 100
    sp[0].sample_index = 3
    with pytest.raises(lib.GrooveError, match="sample_index"):
        E.Sampler(gpu_ctx, pcm, descs, sp)
    # a failed call leaves the context usable
    s.handle_midi_events(T.note_events([(0, 60, True)]))
    s.generate_batch_values(big, 64)
    assert big.download(64)[:, :, 0].any()
    # mixing blocks of different lane counts into one bus is fine; block sums need matching lanes or a 1-lane sink
    bus = gpu_ctx.bus(64)
    gpu_ctx.mix([big, small], 64, bus)
    assert L.groove_block_accumulate(small.h, big.h, 16, 0) != 0
    one = gpu_ctx.block(1, 64)
    assert L.groove_block_accumulate(one.h, big.h, 64, 0) == 0
    assert np.allclose(one.download(64)[:, :, 0], big.download(64).sum(axis=2), atol=1e-6)
    for x in (s, d, small, big, one, bus):
        x.destroy()


def test_sample_rate_change_rebuilds_effects(oracle):
    """Configurable::update_sample_rate reaches effects: ring lengths and filter coefficients follow the new rate."""
    from groove_amd import entities as E
    ctx = E.Context(0)
    fx = E.Effect(ctx, T.FX_DELAY, (T.FxParams * 4)(*[T.fx_params(delay_seconds=0.01)] * 4))
    lp = E.Effect(ctx, T.FX_BIQUAD_LP12, (T.FxParams * 4)(*[T.fx_params(cutoff_hz=2000.0, q=0.9)] * 4))
    ctx.update_sample_rate(48000)
    x = np.zeros((2, 1024, 4), dtype=np.float32)
    x[:, 0, :] = 1.0
    b = ctx.block(4, 256)
    outs, lps = [], []
    b2 = ctx.block(4, 256)
    for i in range(4):
        b.upload(x[:, i * 256:(i + 1) * 256]); fx.transform_audio(b, 256); outs.append(b.download(256))
        b2.upload(x[:, i * 256:(i + 1) * 256]); lp.transform_audio(b2, 256); lps.append(b2.download(256))
    y = np.concatenate(outs, axis=1)
    assert y[0, 480, 0] == 1.0 and np.count_nonzero(y) == 8        # 0.01 s * 48,000 = 480 frames, both channels, 4 lanes
    want = oracle.Fx(T.FX_BIQUAD_LP12, (T.FxParams * 4)(*[T.fx_params(cutoff_hz=2000.0, q=0.9)] * 4), sr=48000).process(x.astype(np.float64))
    assert np.max(np.abs(np.concatenate(lps, axis=1) - want)) <= 2e-6
    ctx.close()


def test_sampler_descriptors_that_would_index_outside_the_bank(gpu_ctx):
    """An empty sample (the fetch clamps to length - 1) and an offset whose sum with the length wraps
    around 2^64 are refused at creation; the asynchronous render checks its arguments like the plain one."""
    from groove_amd import entities as E
    L, h = gpu_ctx.L, gpu_ctx.h
    pcm = np.linspace(-1, 1, 1000, dtype=np.float32)
    sp = (T.SamplerParams * 2)()
    out = C.c_void_p()

    def create(offset, length, frames=1000):
        d = (T.SampleDesc * 1)()
        d[0].offset, d[0].length, d[0].root_hz = offset, length, 0.0
        return L.groove_sampler_create(h, pcm.ctypes.data_as(C.POINTER(C.c_float)), frames, d, 1, sp, 2, C.byref(out))

    assert create(0, 0) != 0 and b"empty sample" in L.groove_last_error(h)
    assert create(2 ** 64 - 50, 100) != 0 and b"exceeds" in L.groove_last_error(h)
    assert create(1001, 0) != 0
    assert create(0, 100, frames=0) != 0
    assert create(0, 1000) == 0
    L.groove_bank_destroy(out)
    s = E.WelshSynth(gpu_ctx, P.welsh_voices(8))
    small, big = gpu_ctx.block(4, 256), gpu_ctx.block(8, 64)
    assert L.groove_bank_render_async(s.h, 16, small.h) != 0 and b"lanes" in L.groove_last_error(h)
    assert L.groove_bank_render_async(s.h, 65, big.h) != 0 and b"capacity" in L.groove_last_error(h)
    assert L.groove_bank_render_async(None, 16, big.h) != 0
    assert L.groove_block_acquire(None) != 0
    s.generate_batch_values_async(big, 64)
    assert np.isfinite(big.download(64)).all()
    for x in (s, small, big):
        x.destroy()


def test_delay_line_geometry_from_hostile_parameters(gpu_ctx):
    """NaN / negative / absurd delay times and tap counts: a one-frame line, a refused effect, or an
    allocation error — never a zero-length ring (the kernels take indices modulo the length) or an
    unbounded tap loop on the device."""
    from groove_amd import entities as E
    L, h = gpu_ctx.L, gpu_ctx.h
    out = C.c_void_p()
    with pytest.raises(lib.GrooveError):  # NaN is not equal to itself: refused by the uniformity check
        E.Effect(gpu_ctx, T.FX_DELAY, (T.FxParams * 2)(*[T.fx_params(delay_seconds=float("nan"))] * 2))
    nan1 = E.Effect(gpu_ctx, T.FX_DELAY, (T.FxParams * 1)(T.fx_params(delay_seconds=float("nan"))))  # one lane: a one-frame line
    nan1.destroy()
    for secs in (-1.0, 0.0):
        fx = E.Effect(gpu_ctx, T.FX_DELAY, (T.FxParams * 2)(*[T.fx_params(delay_seconds=secs)] * 2))
        blk = gpu_ctx.block(2, 64)
        x = np.random.default_rng(0).standard_normal((2, 64, 2)).astype(np.float32)
        blk.upload(x)
        fx.transform_audio(blk, 64)
        y = blk.download(64)
        assert np.array_equal(y[:, 1:, :], x[:, :-1, :])   # one frame of delay
        fx.destroy(); blk.destroy()
    assert L.groove_fx_create(h, T.FX_DELAY, (T.FxParams * 2)(*[T.fx_params(delay_seconds=1e30)] * 2), 2, C.byref(out)) != 0
    assert L.groove_fx_create(h, T.FX_CHORUS, (T.FxParams * 2)(*[T.fx_params(voices=4_000_000_000, delay_seconds=0.01)] * 2), 2, C.byref(out)) != 0
    assert b"voices" in L.groove_last_error(h)


def test_synchronize_deadline_names_the_blocked_stream():
    """A kernel that does not complete must come back as an ERROR, not as a hang: a library stream
    is blocked on purpose (groove_debug_spin: one idle kernel), groove_synchronize and a download return non-zero within
    the deadline and name the stream; nothing is cancelled — a later wait sees the work complete and the ctx works on."""
    import time
    from groove_amd import entities as E
    ctx = E.Context(0)
    try:
        assert ctx.sync_timeout_ms == 60000
        ctx.sync_timeout_ms = 250
        ctx.debug_spin(1, 1500)            # kind stream 1 busy for 1.5 s
        t0 = time.time()
        with pytest.raises(lib.GrooveError, match="groove_synchronize: not complete after 250 ms.*kind stream 1"):  # (safe layout: the same stream object)
            ctx.synchronize()
        assert time.time() - t0 < 1.2
        ctx.sync_timeout_ms = 20000
        ctx.synchronize()                  # the same wait, long enough: completes
        # the ctx stream itself: a blocking copy hits the deadline too
        bus = ctx.bus(256)
        ctx.sync_timeout_ms = 200
        ctx.debug_spin(-1, 1200)
        with pytest.raises(lib.GrooveError, match="copy on the ctx stream: not complete after 200 ms.*ctx stream"):
            bus.download()
        ctx.sync_timeout_ms = 0            # wait for ever: plain hipStreamSynchronize
        ctx.synchronize()
        assert not bus.download().any()
        # still a working context
        ctx.sync_timeout_ms = 20000
        n = 64
        synth = E.WelshSynth(ctx, P.welsh_voices(n))
        synth.handle_midi_events(P.note_on_all(n))
        synth.render_mix(bus, 256)
        assert np.abs(bus.download()).max() > 1e-3
        info = ctx.debug_info()
        import os
        safe = os.environ.get("GROOVE_SAFE_STREAMS") == "1"   # (the whole suite is also run once under the safe layout)
        assert info["streams_created"] == (4 if safe else 8) and info["placeholder_fifth"] is (not safe) and info["comm_before_streams"] is False
        # the library's own host waits are counted (tools/host_blocked.py reads them): this context waited at least for the spin above
        assert info["host_waits"] >= 1 and info["host_waits_blocked"] >= 1 and info["host_wait_ms"] > 1.0 and info["zero_segments"] == 0
    finally:
        ctx.close()


def test_init_with_communicator_first_and_safe_stream_layout():
    """groove_init_comm (the rank's RCCL communicator before the library's streams) on one rank, and the conservative
    stream layout (GROOVE_SAFE_STREAMS=1: one priority, four streams): same bus bits as the default layout for a project
    that uses every bank stream (mixed Welsh / FM / sampler, fused, note events in most blocks)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys, json, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from groove_amd import entities as E, projects as PJ\n"
        "uid = E.Context.new_comm_unique_id()\n"
        "ctx = E.Context(0, comm=(uid, 0, 1))\n"
        "proj = PJ.Project(ctx, 'mixed-131072', np.arange(8192, dtype=np.int64), bank_scale=0.05)\n"
        "bus = ctx.bus(10 * 256)\n"
        "[proj.step(bus, b * 256) for b in range(10)]\n"
        "ctx.bus_reduce(bus, 10 * 256, 0)\n"
        "out = bus.download()\n"
        "print(json.dumps({'sha': hashlib.sha256(out.tobytes()).hexdigest(), 'peak': float(np.abs(out).max()), 'ranks': ctx.comm_ranks(), 'info': ctx.debug_info()}))\n"
        "proj.destroy(); ctx.close()\n" % repo)
    res = {}
    for safe in ("0", "1"):
        env = dict(os.environ, GROOVE_SAFE_STREAMS=safe, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[safe] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["0"]["peak"] > 1.0 and res["0"]["sha"] == res["1"]["sha"]
    assert res["0"]["ranks"] == 1 and res["0"]["info"]["comm_before_streams"] is True
    assert res["0"]["info"]["streams_created"] == 8 and res["1"]["info"]["streams_created"] == 4
    assert res["1"]["info"]["layout"].startswith("safe") and res["1"]["info"]["placeholder_fifth"] is False
