"""GPU: groove_bank_render_async (render-ahead on the library's side streams) produces exactly what
groove_bank_render does, whatever the host interleaves with it on the ctx stream."""
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

FRAMES = 256


def _twin(make):
    return make(), make()


def _render_ahead(gpu_ctx, sync_inst, async_inst, n, blocks, events, fx_sync=(), fx_async=(), frames_of=lambda b: FRAMES, rotation=2,
                  head_async=None, fused_head=False):
    """Walk `blocks` blocks twice: instrument A the plain way (render, effects, mix into bus A);
    instrument B software-pipelined (render of block b+1 submitted before the effects of block b,
    two blocks alternating).  events(b) -> note events applied before block b.  Returns both buses
    and the per-block downloads of the effect outputs."""
    blk_s = gpu_ctx.block(n, FRAMES)
    blk_a = [gpu_ctx.block(n, FRAMES) for _ in range(rotation)]  # rotation 3: blocks are released after their mix (groove_block_release)
    bus_s, bus_a = gpu_ctx.bus(blocks * FRAMES), gpu_ctx.bus(blocks * FRAMES)
    from groove_amd import entities as E
    outs_s, outs_a = [], []
    at = 0
    # plain walk
    for b in range(blocks):
        f = frames_of(b)
        ev = events(b)
        if ev is not None:
            sync_inst.handle_midi_events(ev)
        sync_inst.generate_batch_values(blk_s, f)
        for e in fx_sync:
            e.transform_audio(blk_s, f)
        gpu_ctx.mix([blk_s], f, E._Slice(bus_s, at), accumulate=False)
        outs_s.append(blk_s.download(f))
        at += f
    # render-ahead walk
    at = 0
    ev = events(0)
    if ev is not None:
        async_inst.handle_midi_events(ev)
    head = {}  # block index in the rotation -> stages of the chain already processed behind the render
    if fused_head:    # groove_bank_render_chain_async: the render with the chain's IIR head fused into / queued behind it
        head[0] = async_inst.generate_batch_values_chain_async(blk_a[0], list(fx_async), frames_of(0))
        head_async.append(head[0])
    else:
        async_inst.generate_batch_values_async(blk_a[0], frames_of(0))
    if head_async is not None and not fused_head:
        head[0] = gpu_ctx.transform_chain_async(list(fx_async), blk_a[0], frames_of(0))
        head_async.append(head[0])
    for b in range(blocks):
        cur, nxt = blk_a[b % rotation], blk_a[(b + 1) % rotation]
        f = frames_of(b)
        if b + 1 < blocks:
            ev = events(b + 1)
            if ev is not None:
                async_inst.handle_midi_events(ev)
            if fused_head:
                head[(b + 1) % rotation] = async_inst.generate_batch_values_chain_async(nxt, list(fx_async), frames_of(b + 1))
                head_async.append(head[(b + 1) % rotation])
            else:
                async_inst.generate_batch_values_async(nxt, frames_of(b + 1))
            if head_async is not None and not fused_head:  # groove_fx_chain_process_async: the chain's IIR head right behind the render, on its stream
                head[(b + 1) % rotation] = gpu_ctx.transform_chain_async(list(fx_async), nxt, frames_of(b + 1))
                head_async.append(head[(b + 1) % rotation])
        if head_async is not None:
            gpu_ctx.transform_chain(list(fx_async)[head.pop(b % rotation, 0):], cur, f)
        else:
            for e in fx_async:
                e.transform_audio(cur, f)
        gpu_ctx.mix([cur], f, E._Slice(bus_a, at), accumulate=False)
        if rotation > 2:
            cur.release()
        if b % 3 == 0:  # downloads synchronise the ctx stream: do it on some blocks only, so others stay overlapped
            outs_a.append((b, cur.download(f)))
        at += f
    gpu_ctx.synchronize()
    got_s, got_a = bus_s.download(), bus_a.download()
    for b, o in outs_a:
        if fused_head:  # the fused filter's start states agree with the separate kernel's to f64 rounding, not bit for bit
            err = float(np.max(np.abs(o.astype(np.float64) - outs_s[b])))
            assert err <= 2e-6 * max(1.0, float(np.abs(outs_s[b]).max())), f"block {b}: {err:.3e}"
        else:
            assert np.array_equal(o, outs_s[b]), f"block {b}: render-ahead block differs from the plain walk"
    if fused_head:
        assert np.max(np.abs(got_s.astype(np.float64) - got_a)) <= 2e-6 * n * max(1.0, float(np.abs(outs_s[0]).max()))
    else:
        assert np.array_equal(got_s, got_a), "bus of the render-ahead walk differs from the plain walk"
    assert np.abs(got_s).max() > 1e-3
    for x in (blk_s, *blk_a, bus_s, bus_a):
        x.destroy()
    return got_s


def _short_chain(n):
    """The config-#3 chain with delay times short enough that sound comes out within a few blocks."""
    import ctypes as C

    def arr(**kw):
        a = (T.FxParams * n)()
        raw = np.frombuffer(bytes(bytearray(T.fx_params(**kw))), dtype=np.uint8)
        C.memmove(a, np.tile(raw, n).tobytes(), n * C.sizeof(T.FxParams))
        return a
    lp = arr(q=0.707)
    for i in range(n):
        lp[i].cutoff_hz = 1000.0 + 50.0 * (i % 64)
    return [(T.FX_BIQUAD_LP12, lp), (T.FX_CHORUS, arr(voices=4, delay_seconds=0.02)),
            (T.FX_DELAY, arr(delay_seconds=0.004)), (T.FX_REVERB, arr(attenuation=0.95, reverb_seconds=0.3))]


def test_welsh_chain_render_ahead_is_identical(gpu_ctx):
    """Config #3 shape: grouped Welsh bank + BiQuad -> Chorus -> Delay -> Reverb per voice."""
    from groove_amd import entities as E
    n, blocks = 1024, 14
    params, idx = P.welsh_voices_grouped(n, 0)
    on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    fx_a = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    fx_b = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    _render_ahead(gpu_ctx, a, b, n, blocks, lambda k: on if k == 0 else (off if k == 8 else None), fx_a, fx_b)
    for x in (a, b, *fx_a, *fx_b):
        x.destroy()


@pytest.mark.parametrize("n", [1024, 5])
def test_chain_head_fused_into_the_render(gpu_ctx, kernel_form, n):
    """groove_bank_render_chain_async: behind a bank that renders time-parallel the chain's leading BiQuad is applied INSIDE
    the render kernel (welsh_tp.h HEAD_BQ); behind the serial kernels it rides as its own launch.  Either way one stage is
    taken, and blocks and bus agree with effect-by-effect processing on the ctx stream (to f64 rounding of the filter's
    start states when fused) — through a note-off, ragged blocks and a reset."""
    from groove_amd import entities as E
    blocks = 14
    params, idx = P.welsh_voices_grouped(n, 0)
    on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    fx_a = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    fx_b = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    sizes = [256, 256, 100, 256, 7, 256, 255, 256, 256, 256, 64, 256, 256, 256]
    assert ("time-parallel" in b.kernel_form(256, False)) == (kernel_form == "time-parallel")
    for rep in range(2):
        taken = []
        _render_ahead(gpu_ctx, a, b, n, blocks, lambda k: on if k == 0 else (off if k == 8 else None), fx_a, fx_b,
                      frames_of=lambda k: sizes[k], rotation=3, head_async=taken, fused_head=True)
        assert taken and all(t == 1 for t in taken), taken
        for x in (a, b, *fx_a, *fx_b):
            x.reset()
    for x in (a, b, *fx_a, *fx_b):
        x.destroy()


@pytest.mark.parametrize("rotation", [2, 3])
def test_chain_head_behind_the_render_is_identical(gpu_ctx, rotation):
    """groove_fx_chain_process_async: the chain's leading BiQuad processed on the render's side stream, one block ahead of
    the ctx stream's walk, the rest of the chain (fused run, all-passes, row sums for the mix) on the ctx stream: same
    blocks and same bus, bit for bit, as effect by effect on the ctx stream — through a note-off, ragged blocks, a reset."""
    from groove_amd import entities as E
    n, blocks = 1024, 14
    params, idx = P.welsh_voices_grouped(n, 0)
    on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    fx_a = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    fx_b = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    sizes = [256, 256, 100, 256, 7, 256, 255, 256, 256, 256, 64, 256, 256, 256]
    for rep in range(2):
        taken = []
        _render_ahead(gpu_ctx, a, b, n, blocks, lambda k: on if k == 0 else (off if k == 8 else None), fx_a, fx_b,
                      frames_of=lambda k: sizes[k], rotation=rotation, head_async=taken)
        assert taken and all(t == 1 for t in taken), taken   # the BiQuad, and only it, rides behind the render
        for x in (a, b, *fx_a, *fx_b):
            x.reset()
    # a block with no pending render: nothing is taken, and the whole chain still works on the ctx stream
    blk = gpu_ctx.block(n, FRAMES)
    assert gpu_ctx.transform_chain_async(fx_b, blk, FRAMES) == 0
    b.generate_batch_values_async(blk, FRAMES)
    assert gpu_ctx.transform_chain_async(fx_b[1:], blk, FRAMES) == 0   # starts with a stage of the other sort
    gpu_ctx.transform_chain(fx_b, blk, FRAMES)
    blk.destroy()
    for x in (a, b, *fx_a, *fx_b):
        x.destroy()


@pytest.mark.parametrize("n", [1024, 8192])
def test_three_block_rotation_with_release_is_identical(gpu_ctx, n):
    """Blocks released after their mix (groove_block_release) and reused two steps later: the render
    into a released block waits for that release only.  Grouped bank and (8192 interleaved voices) a
    regrouped one; a download right after a release, and a note-off in the middle, take the block
    back to the conservative path."""
    from groove_amd import entities as E
    blocks = 15
    if n == 1024:
        params, idx = P.welsh_voices_grouped(n, 0)
        on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    else:
        params = P.welsh_voices(n)
        on, off = P.note_on_all(n), P.note_off_all(n)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    fx_a = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    fx_b = [E.Effect(gpu_ctx, k, p) for k, p in _short_chain(n)]
    _render_ahead(gpu_ctx, a, b, n, blocks, lambda k: on if k == 0 else (off if k == 9 else None), fx_a, fx_b, rotation=3)
    for x in (a, b, *fx_a, *fx_b):
        x.destroy()


def test_regrouped_welsh_render_ahead_is_identical(gpu_ctx):
    """Interleaved patches: the bank is regrouped inside the library, so the asynchronous render goes
    through the scratch block and the gather kernel; ragged block lengths."""
    from groove_amd import entities as E
    n, blocks = 8192, 9
    params = P.welsh_voices(n)
    on, off = P.note_on_all(n), P.note_off_all(n)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    lens = [256, 100, 7, 256, 1, 255, 64, 256, 33]
    _render_ahead(gpu_ctx, a, b, n, blocks, lambda k: on if k == 0 else (off if k == 5 else None), frames_of=lambda k: lens[k])
    a.destroy(); b.destroy()


def test_fm_and_sampler_render_ahead_is_identical(gpu_ctx):
    from groove_amd import entities as E
    n = 512
    fm = P.fm_voices(n)
    a, b = _twin(lambda: E.FmSynth(gpu_ctx, fm))
    _render_ahead(gpu_ctx, a, b, n, 8, lambda k: P.note_on_all(n) if k == 0 else (P.note_off_all(n) if k == 4 else None))
    a.destroy(); b.destroy()
    pcm, descs, _ = P.drum_bank(scale=0.05)
    sp = P.sampler_voices(n)
    keys = P.sampler_keys(n)
    a, b = _twin(lambda: E.Sampler(gpu_ctx, pcm, descs, sp))
    ev = T.note_events_np(np.arange(n, dtype=np.uint32), keys, True)
    _render_ahead(gpu_ctx, a, b, n, 6, lambda k: ev if k in (0, 3) else None)
    a.destroy(); b.destroy()


def test_async_render_then_fused_mix_and_state(gpu_ctx):
    """Switching between the asynchronous render, the fused render+mix and the state download on the
    same bank keeps the order of calls (different side streams are joined when the bank changes sets)."""
    from groove_amd import entities as E
    n = 2048
    params, idx = P.welsh_voices_grouped(n, 0)
    on = P.grouped_note_events(idx, True)
    a, b = _twin(lambda: E.WelshSynth(gpu_ctx, params))
    blk_a, blk_b = gpu_ctx.block(n, FRAMES), gpu_ctx.block(n, FRAMES)
    bus_a, bus_b = gpu_ctx.bus(FRAMES), gpu_ctx.bus(FRAMES)
    a.handle_midi_events(on); b.handle_midi_events(on)
    for step in range(6):
        a.generate_batch_values(blk_a, FRAMES)
        b.generate_batch_values_async(blk_b, FRAMES)
        a.render_mix(bus_a, FRAMES); b.render_mix(bus_b, FRAMES)
        assert np.array_equal(blk_a.download(FRAMES), blk_b.download(FRAMES)), f"step {step}: blocks differ"
        assert np.array_equal(bus_a.download(), bus_b.download()), f"step {step}: fused buses differ"
    assert np.array_equal(a.download_state(), b.download_state())
    for x in (a, b, blk_a, blk_b, bus_a, bus_b):
        x.destroy()
