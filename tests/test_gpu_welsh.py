"""GPU parity, Welsh voices: HIP path through the
C ABI vs the f64 oracle on the same seeded inputs.

Tolerances (fp32 device output vs f64 oracle):
  per voice:  RMS error <= 1e-5 (signal RMS ~0.1-0.7), i.e. the north-star bar per voice;
  bus / V:    RMS error <= 1e-5 (north star), expected ~1e-7.
"""
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

TOL_RMS = 1e-5


def _run_pair(gpu_ctx, oracle, n, blocks, off_block, frames=256, first_voice=0):
    from groove_amd import entities as E
    params = P.welsh_voices(n, first_voice)
    synth = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, frames)
    ob = oracle.Bank.welsh(params)
    on, off = P.note_on_all(n, first_voice), P.note_off_all(n, first_voice)
    got, want = [], []
    for b in range(blocks):
        if b == 0:
            synth.handle_midi_events(on); ob.note_events(on)
        if b == off_block:
            synth.handle_midi_events(off); ob.note_events(off)
        synth.generate_batch_values(block, frames)
        got.append(block.download(frames))
        want.append(ob.render(frames))
    synth.destroy(); block.destroy()
    return np.concatenate(got, axis=1).astype(np.float64), np.concatenate(want, axis=1)


def test_welsh_per_voice_parity_32_patches_full_render(gpu_ctx, oracle):
    """Config #2 timing: 172 blocks x 256 frames, note-off at frame 22,016, all 32 patches."""
    got, want = _run_pair(gpu_ctx, oracle, 32, P.RENDER_BLOCKS, P.NOTE_OFF_FRAME // 256)
    assert np.isfinite(got).all()
    err = got - want
    for v in range(32):
        rms = np.sqrt(np.mean(err[:, :, v] ** 2))
        sig = np.sqrt(np.mean(want[:, :, v] ** 2))
        assert sig > 1e-3, f"voice {v} is silent"
        assert rms <= TOL_RMS, f"voice {v}: rms error {rms:.3e} (signal {sig:.3e})"


def test_welsh_config2_bus_parity_256_voices(gpu_ctx, oracle):
    """BASELINE config #2: 256 Welsh voices, 44,032 frames; mix bus via groove_mix."""
    from groove_amd import entities as E
    n, frames = 256, 256
    params = P.welsh_voices(n)
    synth = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, frames)
    bus = gpu_ctx.bus(P.RENDER_BLOCKS * frames)
    ob = oracle.Bank.welsh(params)
    on, off = P.note_on_all(n), P.note_off_all(n)
    want = []
    for b in range(P.RENDER_BLOCKS):
        if b == 0:
            synth.handle_midi_events(on); ob.note_events(on)
        if b == P.NOTE_OFF_FRAME // frames:
            synth.handle_midi_events(off); ob.note_events(off)
        synth.generate_batch_values(block, frames)
        gpu_ctx.mix([block], frames, E._Slice(bus, b * frames))
        want.append(ob.render_bus(frames))
    got = bus.download().astype(np.float64) / n
    want = np.concatenate(want, axis=0) / n
    rms = np.sqrt(np.mean((got - want) ** 2))
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    assert rms <= TOL_RMS, f"bus rms error {rms:.3e}"
    synth.destroy(); block.destroy(); bus.destroy()


def test_welsh_partial_blocks_and_ragged_voice_count(gpu_ctx, oracle):
    """n not a multiple of 64 / 256, block lengths 1, 63, 256, 100: same stream as one long render."""
    from groove_amd import entities as E
    n = 77
    params = P.welsh_voices(n, first_voice=5)
    synth = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, 256)
    ob = oracle.Bank.welsh(params)
    on = P.note_on_all(n, 5)
    synth.handle_midi_events(on); ob.note_events(on)
    got, want = [], []
    for frames in (1, 63, 256, 100, 7):
        synth.generate_batch_values(block, frames)
        got.append(block.download(frames))
        want.append(ob.render(frames))
    err = np.concatenate(got, axis=1).astype(np.float64) - np.concatenate(want, axis=1)
    assert np.sqrt(np.mean(err ** 2)) <= TOL_RMS
    synth.destroy(); block.destroy()


def test_welsh_idle_voices_are_silent_and_noise_is_bit_exact(gpu_ctx, oracle):
    from groove_amd import entities as E
    n = 64
    params = P.welsh_voices(n)
    for i in range(n):  # osc1 = noise only, filter wide open and static, no LFO
        p = params[i]
        p.oscillator_1.waveform = T.WAVE_NOISE
        p.oscillator_2.waveform = T.WAVE_NONE
        p.oscillator_mix = 1.0
        p.lfo_routing = T.LFO_NONE
        p.filter_cutoff_end = 0.0
    synth = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, 256)
    synth.generate_batch_values(block, 256)
    assert not block.download(256).any(), "voices must be silent before any note-on"
    ev = T.note_events([(v, 60, True) for v in range(0, n, 2)])
    synth.handle_midi_events(ev)
    synth.generate_batch_values(block, 256)
    out = block.download(256)
    assert not out[:, :, 1::2].any(), "un-triggered voices stay silent"
    assert out[:, :, 0::2].any()
    # integer noise generator state must match the oracle's sequence exactly after 256 ticks
    st = synth.download_state()
    x1, x2 = np.uint32(0x70f4f854), np.uint32(0xe1e9f0a7)
    with np.errstate(over="ignore"):
        for _ in range(256):
            x1 = x1 ^ x2
            x2 = np.uint32(x2 + x1)
    # WelshState layout: o1 = {phase u64, x1, x2, flags, pad} → words 2, 3
    assert int(st[2, 0]) == int(x1) and int(st[3, 0]) == int(x2)
    synth.destroy(); block.destroy()


def test_fused_render_mix_matches_materialised_and_oracle(gpu_ctx, oracle):
    """groove_bank_render_mix (DPP wave sums + partial rows, no voice block in HBM) against the
    oracle bus and against render + groove_mix on an identical second bank.  Grouped layout
    (patch-uniform wavefronts → scalar-parameter kernel) and interleaved layout (generic kernel),
    ragged voice counts, partial blocks."""
    from groove_amd import entities as E
    for grouped, n in ((True, 64 * 32 + 40), (False, 300)):
        if grouped:
            params, idx = P.welsh_voices_grouped(n)
            on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
        else:
            params = P.welsh_voices(n)
            on, off = P.note_on_all(n), P.note_off_all(n)
        a, b = E.WelshSynth(gpu_ctx, params), E.WelshSynth(gpu_ctx, params)
        ob = oracle.Bank.welsh(params)
        block = gpu_ctx.block(n, 256)
        sizes = [256] * 6 + [100, 1, 63, 256, 192]
        total = sum(sizes)
        bus_a, bus_b = gpu_ctx.bus(total), gpu_ctx.bus(total)
        want = []
        pos = 0
        for k, fr in enumerate(sizes):
            if k == 0:
                a.handle_midi_events(on); b.handle_midi_events(on); ob.note_events(on)
            if k == 4:
                a.handle_midi_events(off); b.handle_midi_events(off); ob.note_events(off)
            a.render_mix(bus_a, fr, at_frame=pos)
            b.generate_batch_values(block, fr)
            gpu_ctx.mix([block], fr, E._Slice(bus_b, pos))
            want.append(ob.render_bus(fr))
            pos += fr
        ga, gb = bus_a.download().astype(np.float64) / n, bus_b.download().astype(np.float64) / n
        want = np.concatenate(want, axis=0) / n
        assert np.sqrt(np.mean(want ** 2)) > 1e-3
        assert np.sqrt(np.mean((ga - want) ** 2)) <= 1e-6, "fused vs oracle"
        assert np.max(np.abs(ga - gb)) <= 2e-6, "fused vs materialised"
        # accumulate = 1 adds onto the bus
        a2 = E.WelshSynth(gpu_ctx, params)
        a2.handle_midi_events(on)
        bus_c = gpu_ctx.bus(256)
        a2.render_mix(bus_c, 256)
        first = bus_c.download().copy()
        a2.render_mix(bus_c, 256, accumulate=True)
        assert np.abs(bus_c.download()).sum() > np.abs(first).sum() * 0 and np.isfinite(bus_c.download()).all()
        for x in (a, b, a2):
            x.destroy()
        block.destroy()


def test_uniform_and_generic_kernels_agree(gpu_ctx):
    """The same voices laid out grouped (scalar-parameter kernel) and interleaved (per-lane
    kernel) must produce the same per-voice blocks: identical arithmetic, different dispatch."""
    from groove_amd import entities as E
    n = 64 * 32
    pg, idx = P.welsh_voices_grouped(n)
    pi = P.welsh_voices(n)
    g, i = E.WelshSynth(gpu_ctx, pg), E.WelshSynth(gpu_ctx, pi)
    g.handle_midi_events(P.grouped_note_events(idx, True)); i.handle_midi_events(P.note_on_all(n))
    bg, bi = gpu_ctx.block(n, 256), gpu_ctx.block(n, 256)
    for _ in range(12):
        g.generate_batch_values(bg, 256); i.generate_batch_values(bi, 256)
        og, oi = bg.download(256), bi.download(256)
        assert np.max(np.abs(og - oi[:, :, idx])) <= 1e-6
    g.destroy(); i.destroy(); bg.destroy(); bi.destroy()


def test_set_param_and_sample_rate(gpu_ctx, oracle):
    """Controllable (dca gain / pan / cutoff) and Configurable::update_sample_rate."""
    from groove_amd import entities as E
    n = 128
    params = P.welsh_voices(n)
    s = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(n, 256)
    s.control_set_param_by_index(T.CTL_WELSH_DCA_GAIN, 0.5)
    s.control_set_param_by_index(T.CTL_WELSH_DCA_PAN, 0.5, voice=3)   # centre
    for i in range(n):
        params[i].dca_gain = 0.5
    params[3].dca_pan = 0.0
    ob = oracle.Bank.welsh(params)
    on = P.note_on_all(n)
    s.handle_midi_events(on); ob.note_events(on)
    s.generate_batch_values(block, 256)
    err = block.download(256).astype(np.float64) - ob.render(256)
    assert np.sqrt(np.mean(err ** 2)) <= 1e-6
    s.destroy()
    # a second context at 22.05 kHz: envelopes, phases and filters re-derived for the new rate;
    # cutoffs above 0.49 SR exercise the Nyquist clamp.  (Not 24 kHz: A notes are exact
    # sub-multiples there, a square edge then lands EXACTLY on a frame and the f64 oracle's
    # accumulated rounding, not the algorithm, decides the side.)
    ctx2 = E.Context(0)
    ctx2.update_sample_rate(22050)
    assert ctx2.sample_rate == 22050
    params = P.welsh_voices(32)
    s2 = E.WelshSynth(ctx2, params)
    ob2 = oracle.Bank.welsh(params, sr=22050)
    on = P.note_on_all(32)
    s2.handle_midi_events(on); ob2.note_events(on)
    b2 = ctx2.block(32, 256)
    got, want = [], []
    for _ in range(8):
        s2.generate_batch_values(b2, 256)
        got.append(b2.download(256)); want.append(ob2.render(256))
    err = np.concatenate(got, axis=1) - np.concatenate(want, axis=1)
    assert np.sqrt(np.mean(err ** 2)) <= 1e-5
    ctx2.close()


def test_rccl_single_rank_bus_reduce(gpu_ctx):
    """groove_comm_* + groove_bus_reduce with world_size 1 (the 1-GPU box): RCCL loads, the
    communicator initialises, and the reduce leaves the bus intact."""
    uid = gpu_ctx.comm_unique_id()
    assert len(uid) == 128
    gpu_ctx.comm_init(uid, 0, 1)
    bus = gpu_ctx.bus(512)
    import ctypes as C
    from groove_amd import lib
    vals = np.arange(1024, dtype=np.float32)
    lib.check(gpu_ctx.L.groove_upload(gpu_ctx.h, bus.ptr, vals.ctypes.data_as(C.POINTER(C.c_float)), vals.size), gpu_ctx.h)
    gpu_ctx.bus_reduce(bus, 512, 0)
    gpu_ctx.synchronize()
    assert np.array_equal(bus.download().reshape(-1), vals)
    with pytest.raises(lib.GrooveError, match="already has a communicator"):
        gpu_ctx.comm_init(uid, 0, 1)
    assert gpu_ctx.L.groove_comm_destroy(gpu_ctx.h) == 0   # (the session's ctx goes on without one)
    assert gpu_ctx.comm_ranks() == 1
    bus.destroy()


def test_pipelined_blocks_with_events_controls_and_state_reads(oracle, monkeypatch):
    """The block pipeline of the fused path (render_mix_pipelined: every base kind on its own stream,
    block b+1's kernels not waiting for block b's bus reduction) forced onto a small grouped bank
    (GROOVE_PIPELINE_MIN_WAVES=1), with everything that has to join the pipeline in between: note
    events mid-render, a control change, a state download, a materialised render.  Bus vs the oracle."""
    from groove_amd import entities as E
    monkeypatch.setenv("GROOVE_PIPELINE_MIN_WAVES", "1")
    monkeypatch.setenv("GROOVE_TP_MAX_VOICES", "0")   # the serial kernels' pipeline is the subject here
    ctx = E.Context(0)
    n, frames, blocks = 2048, 256, 24
    params, vidx = P.welsh_voices_grouped(n)
    on, off = P.grouped_note_events(vidx, True), P.grouped_note_events(vidx, False)
    synth = E.WelshSynth(ctx, params)
    ob = oracle.Bank.welsh(params)
    bus = ctx.bus(blocks * frames)
    block = ctx.block(n, frames)
    want = []
    for b in range(blocks):
        if b == 0:
            synth.handle_midi_events(on); ob.note_events(on)
        if b == 9:
            synth.handle_midi_events(off); ob.note_events(off)
        if b == 13:  # re-trigger half of the voices while the others release
            half = T.note_events_np(np.arange(0, n, 2, dtype=np.uint32), (36 + (7 * vidx[::2]) % 49).astype(np.uint8), True)
            synth.handle_midi_events(half); ob.note_events(half)
        if b == 5:
            state = synth.download_state()  # joins the pipeline, reads consistent state
            assert state.shape[1] == n
        if b == 17:  # a materialised block in the middle of the fused ones
            synth.generate_batch_values(block, frames)
            ctx.mix([block], frames, E._Slice(bus, b * frames))
        else:
            synth.render_mix(bus, frames, at_frame=b * frames)
        want.append(ob.render_bus(frames))
    got = bus.download().astype(np.float64)
    want = np.concatenate(want, axis=0)
    assert np.isfinite(got).all()
    rms = np.sqrt(np.mean(((got - want) / n) ** 2))
    assert rms <= 1e-6, rms
    synth.destroy(); bus.destroy(); block.destroy(); ctx.close()


def test_interleaved_bank_is_regrouped_transparently(gpu_ctx, oracle):
    """Voice i uses patch i mod 32 (config #2's literal rule) in a bank large enough that the library
    regroups it patch-major internally (lane permutation): note events, state download and the
    materialised block must keep the caller's voice order; fused bus and sampled voices vs the oracle."""
    from groove_amd import entities as E
    n, frames, blocks = 8192, 256, 4   # 8192 runs of one voice: too short for the uniform kernels unless regrouped
    params = P.welsh_voices(n)
    synth = E.WelshSynth(gpu_ctx, params)
    on = P.note_on_all(n)
    # only the even voices play, so a wrong event mapping is audible
    ev_on = T.note_events_np(np.arange(0, n, 2, dtype=np.uint32), P.voice_keys(n)[::2], True)
    synth.handle_midi_events(ev_on)
    sample = np.arange(0, n, 37)
    ob = oracle.Bank.welsh((T.WelshParams * len(sample))(*[params[int(i)] for i in sample]))
    ob.note_events(T.note_events_np(np.flatnonzero(sample % 2 == 0).astype(np.uint32), P.voice_keys(n)[sample[sample % 2 == 0]], True))
    block = gpu_ctx.block(n, frames)
    for b in range(blocks):
        synth.generate_batch_values(block, frames)
        got = block.download(frames)[:, :, sample].astype(np.float64)
        want = ob.render(frames)
        assert np.abs(got[:, :, sample % 2 == 1]).max() == 0.0, "an odd voice sounds: note events landed on the wrong lanes"
        rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
        assert rms.max() <= TOL_RMS, rms.max()
    # a control change on ONE voice of the regrouped bank (its run of equal patches splits around it)
    v_ctl = int(sample[2])
    assert v_ctl % 2 == 0
    synth.control_set_param_by_index(T.CTL_WELSH_DCA_PAN, 0.9, voice=v_ctl)
    pm = T.WelshParams.from_buffer_copy(params[v_ctl])  # a copy: params[...] is a view into the table
    pm.dca_pan = 0.9 * 2.0 - 1.0
    oc = oracle.Bank.welsh((T.WelshParams * 1)(pm))
    oc.note_events(T.note_events_np(np.array([0], dtype=np.uint32), P.voice_keys(n)[v_ctl:v_ctl + 1], True))
    for _ in range(blocks):
        oc.render(frames)  # catch up with the blocks already rendered (pan is memoryless: only gains change)
    synth.generate_batch_values(block, frames)
    got = block.download(frames)[:, :, v_ctl].astype(np.float64)
    want = oc.render(frames)[:, :, 0]
    assert np.sqrt(np.mean((got - want) ** 2)) <= TOL_RMS
    ob.render(frames)  # keep the sampled oracle in step
    st = synth.download_state()
    idle_word = st[:, 1::2]
    assert (idle_word == idle_word[:, :1]).all(), "odd (never triggered) voices must all hold the initial state"
    assert not (st[:, 0::2] == st[:, 1:2]).all(), "even voices played: their state moved"
    # fused form of the same bank against a fresh grouped bank of the same voices
    a = E.WelshSynth(gpu_ctx, params); a.handle_midi_events(on)
    pg, vidx = P.welsh_voices_grouped(n)
    g = E.WelshSynth(gpu_ctx, pg); g.handle_midi_events(P.grouped_note_events(vidx, True))
    ba, bg = gpu_ctx.bus(frames), gpu_ctx.bus(frames)
    for b in range(blocks):
        a.render_mix(ba, frames); g.render_mix(bg, frames)
        assert np.max(np.abs(ba.download() - bg.download())) / n <= 1e-6
    for x in (synth, a, g, block, ba, bg):
        x.destroy()


def test_mix_uses_the_blocks_row_sums_only_while_they_are_valid(gpu_ctx, oracle):
    """A block-writing render leaves the block's lane sums behind and groove_mix reduces those rows instead of reading the
    block back; whatever else writes the block (an effect in place, an upload, an accumulate into it, a zero) must make
    the mix read the block again, and so must a mix over another frame count than was rendered."""
    from groove_amd import entities as E
    n = 96
    params = P.welsh_voices(n)
    synth = E.WelshSynth(gpu_ctx, params)
    synth.handle_midi_events(P.note_on_all(n))
    block, other = gpu_ctx.block(n, 256), gpu_ctx.block(n, 256)
    bus = gpu_ctx.bus(256)

    def mixed(frames=256):
        gpu_ctx.mix([block], frames, bus)
        return bus.download(frames).astype(np.float64)

    synth.generate_batch_values(block, 256)
    full = block.download(256).astype(np.float64)
    assert np.abs(full).max() > 0.1
    assert np.max(np.abs(mixed() - full.sum(axis=2).T)) <= 1e-5                      # cached rows
    assert np.max(np.abs(mixed(100) - full[:, :100].sum(axis=2).T)) <= 1e-5          # other frame count: reads the block
    g = E.Effect(gpu_ctx, T.FX_GAIN, (T.FxParams * n)(*[T.fx_params(ceiling=0.5)] * n))
    g.transform_audio(block, 256)
    assert np.max(np.abs(mixed() - 0.5 * full.sum(axis=2).T)) <= 1e-5                # effect in place
    synth.generate_batch_values(block, 256)
    nxt = block.download(256).astype(np.float64)
    half = (0.25 * nxt).astype(np.float32)
    block.upload(half)
    assert np.max(np.abs(mixed() - half.astype(np.float64).sum(axis=2).T)) <= 1e-5   # upload
    synth.generate_batch_values(block, 256)
    cur = block.download(256).astype(np.float64)
    other.upload(half)
    assert gpu_ctx.L.groove_block_accumulate(block.h, other.h, 256, 1) == 0
    assert np.max(np.abs(mixed() - (cur + half).sum(axis=2).T)) <= 1e-5              # accumulate into it
    synth.generate_batch_values(block, 256)
    assert gpu_ctx.L.groove_block_zero(block.h) == 0
    assert not mixed().any()                                                          # zero
    for x in (g, synth, block, other, bus):
        x.destroy()


def test_regrouped_bank_blocks_keep_the_library_order_until_someone_looks(gpu_ctx, oracle):
    """Lazy lane order (groove_hip.hip, groove_block): a render of a regrouped bank leaves the block in the library's
    lane order; the mix needs no other (same bus as after a download, which produces the caller's order), an effect
    with per-lane parameters, an element-wise accumulate and the raw pointer all see the caller's order, and rendering
    again after any of these is consistent."""
    import ctypes as C
    from groove_amd import entities as E, lib
    n, frames = 8192, 256
    params = P.welsh_voices(n)          # voice i: patch i mod 32 -> regrouped patch-major inside the library
    keys = P.voice_keys(n)
    def fresh():
        s_ = E.WelshSynth(gpu_ctx, params)
        s_.handle_midi_events(T.note_events_np(np.arange(0, n, 2, dtype=np.uint32), keys[::2], True))  # even voices only
        return s_
    a, b = fresh(), fresh()
    blk_a, blk_b = gpu_ctx.block(n, frames), gpu_ctx.block(n, frames)
    bus_a, bus_b = gpu_ctx.bus(frames), gpu_ctx.bus(frames)
    gains = np.linspace(0.25, 1.0, n).astype(np.float32)           # a per-lane parameter: lane order matters
    fp = (T.FxParams * n)()
    for i in range(n):
        fp[i] = T.fx_params(ceiling=float(gains[i]))
    fx = E.Effect(gpu_ctx, T.FX_GAIN, fp)
    for it in range(3):
        # A: render, mix straight away (library order, lane sums); B: render, download first (caller's order), then mix
        a.generate_batch_values(blk_a, frames)
        gpu_ctx.mix([blk_a], frames, bus_a)
        b.generate_batch_values(blk_b, frames)
        host_b = blk_b.download(frames)
        gpu_ctx.mix([blk_b], frames, bus_b)
        ga, gb = bus_a.download().astype(np.float64), bus_b.download().astype(np.float64)
        assert np.abs(ga).max() > 1e-2 and np.array_equal(ga, gb), it
        assert np.abs(host_b[:, :, 1::2]).max() == 0.0 and np.abs(host_b[:, :, 0::2]).max() > 1e-3   # the caller's lanes
        want = host_b.astype(np.float64).sum(axis=2).T
        assert np.max(np.abs(ga - want)) <= 2e-6 * max(1.0, np.abs(host_b).sum(axis=2).max())
        host_a = blk_a.download(frames)                               # now A is asked too: same block, bit for bit
        assert np.array_equal(host_a.view(np.uint32), host_b.view(np.uint32))
        # an effect with per-lane parameters on a freshly rendered (library-order) block
        a.generate_batch_values(blk_a, frames)
        b.generate_batch_values(blk_b, frames)
        ref = blk_b.download(frames)
        fx.transform_audio(blk_a, frames)
        got = blk_a.download(frames)
        assert np.array_equal(got, ref * gains[None, None, :])
    # the raw pointer of a library-order block is the caller's order
    a.generate_batch_values(blk_a, frames)
    b.generate_batch_values(blk_b, frames)
    ref = blk_b.download(frames)
    dev = blk_a.device_ptr()
    row = np.empty(n, dtype=np.float32)
    lib.check(gpu_ctx.L.groove_download(gpu_ctx.h, C.c_void_p(dev + 17 * n * 4), row.ctypes.data_as(C.POINTER(C.c_float)), n), gpu_ctx.h)
    assert np.array_equal(row, ref[0, 17, :])
    # element-wise accumulate of two library-order blocks, and the lane sum into a one-lane block
    a.generate_batch_values(blk_a, frames)
    b.generate_batch_values(blk_b, frames)
    one = gpu_ctx.block(1, frames)
    gpu_ctx.L.groove_block_accumulate(one.h, blk_a.h, frames, 0)
    s1 = one.download(frames)[:, :, 0].astype(np.float64)
    hb = blk_b.download(frames).astype(np.float64)                    # (same voices, same state history)
    gpu_ctx.L.groove_block_accumulate(blk_b.h, blk_a.h, frames, 1)
    both = blk_b.download(frames).astype(np.float64)
    ha = blk_a.download(frames).astype(np.float64)
    assert np.max(np.abs(both - (ha + hb))) <= 1e-6
    assert np.max(np.abs(s1 - ha.sum(axis=2))) <= 2e-6 * max(1.0, np.abs(ha).sum(axis=2).max())
    for x in (a, b, fx, blk_a, blk_b, bus_a, bus_b, one):
        x.destroy()
