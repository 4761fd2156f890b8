"""GPU tier: the library-proportioned patch table (groove_amd/patches.py LIBRARY_*: 106 synthetic patches whose categories follow the
reference's 106 shipped patch files — tools/library_proportions.py) and the workload built on it, `welsh-1m-library`.

What it reaches that the 32-patch benchmark table does not: square and sawtooth LFOs on the pitch and the
pulse width (the smooth kinds' recurrences with an exact re-seed on the frame of an LFO edge), filters with ripples of 4.3 - 10.7 under
envelope and LFO sweeps (the two-sided coefficient form, in every kind since round 6), a noise LFO on the pitch and the resonance
routing (the exact-f64 kind, the only two slots left there).

Tolerance: per voice RMS <= 1e-5 of max(1, the voice's level) against the f64 oracle (a ripple of 10.7 rings above full scale);
bus / V <= 1e-6 between forms.
"""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T, projects as PJ

pytestmark = pytest.mark.gpu
KEYS_OF = (43, 66)   # (no A: docs/DSP_SPEC.md section 2)


def _bank():
    S = P.LIBRARY_SLOTS
    pats = [P.library_patch(s) for s in range(S) for _ in KEYS_OF]
    keys = np.array([k for _ in range(S) for k in KEYS_OF], dtype=np.uint8)
    n = len(pats)
    lanes = np.arange(n, dtype=np.uint32)
    return (T.WelshParams * n)(*pats), T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)


@pytest.fixture(params=["all-kinds serial", "role-split", "time-parallel", "per-kind pipelined"])
def form(request, gpu_ctx):
    old = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves
    if request.param != "time-parallel":
        gpu_ctx.time_parallel_max_voices = 0
    if request.param != "role-split":
        gpu_ctx.split_max_waves = 0
    if request.param == "per-kind pipelined":
        gpu_ctx.pipeline_min_waves = 1
    yield request.param
    gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = old


def test_library_table_per_voice_parity(gpu_ctx, oracle, form):
    """Every slot, two keys, 60 blocks with the note-off at 40 (LFO edges at 2.07 - 31.7 Hz fall inside; so do the release and the idle
    tail), the voice block against the oracle voice by voice — and the fused bus of the same bank against the block's sum."""
    from groove_amd import entities as E
    params, on, off = _bank()
    n, frames, blocks = len(params), 256, 60
    synth, fused = E.WelshSynth(gpu_ctx, params), E.WelshSynth(gpu_ctx, params)
    want_form = {"all-kinds serial": "all base kinds in one launch", "role-split": "role-split", "time-parallel": "time-parallel", "per-kind pipelined": "one launch per base kind"}[form]
    assert want_form in fused.kernel_form(frames, True), fused.kernel_form(frames, True)
    block, bus = gpu_ctx.block(n, frames), gpu_ctx.bus(blocks * frames)
    ob = oracle.Bank.welsh(params)
    got, want = [], []
    for b in range(blocks):
        if b == 0:
            synth.handle_midi_events(on); fused.handle_midi_events(on); ob.note_events(on)
        if b == 40:
            synth.handle_midi_events(off); fused.handle_midi_events(off); ob.note_events(off)
        synth.generate_batch_values(block, frames)
        fused.render_mix(bus, frames, at_frame=b * frames)
        got.append(block.download(frames))
        want.append(ob.render(frames))
    got, want = np.concatenate(got, axis=1).astype(np.float64), np.concatenate(want, axis=1)
    assert np.isfinite(got).all()
    level = np.sqrt(np.mean(want ** 2, axis=(0, 1)))
    per_voice = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1))) / np.maximum(1.0, level)
    assert per_voice.max() <= 1e-5, (form, int(per_voice.argmax()) // len(KEYS_OF), per_voice.max())
    assert (level > 1e-3).sum() >= n - 4
    fbus = bus.download().astype(np.float64)
    assert np.max(np.abs(fbus - got.sum(axis=2).T)) / n <= 1e-6, "fused bus and the voice block's sum differ"
    print(f"library table, {form}: worst per-voice RMS {per_voice.max():.2e} (slot {int(per_voice.argmax()) // len(KEYS_OF)})")
    synth.destroy(); fused.destroy(); block.destroy(); bus.destroy()


V = 1_000_000
NOTE_OFF_BLOCK, BLOCKS = 12, 30


def test_library_workload_full_size(gpu_ctx, oracle):
    """welsh-1m-library at BASELINE's size through the kernels bench.py times (one kernel per base kind, blocks pipelined): the fused
    bus is finite and loud; the entity-boundary form's voice block agrees with the oracle on 5 voices per slot at sampled frames of
    every block; the two forms' buses agree; two contiguous shards sum to the whole."""
    import ctypes as C
    from groove_amd import entities as E, lib
    from groove_amd.parallel import voice_range
    spec = PJ.plan("welsh-1m-library", np.arange(V, dtype=np.int64))[0]
    params, voice = spec["params"], spec["voice"]
    keys = (36 + (7 * voice) % 49).astype(np.uint8)
    lanes = np.arange(V, dtype=np.uint32)
    on, off = T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)
    frames, blocks = 256, BLOCKS
    full = E.WelshSynth(gpu_ctx, params)
    assert "one launch per base kind" in full.kernel_form(frames, True)
    bus = gpu_ctx.bus(blocks * frames)
    for b in range(blocks):
        if b == 0:
            full.handle_midi_events(on)
        if b == NOTE_OFF_BLOCK:
            full.handle_midi_events(off)
        full.render_mix(bus, frames, at_frame=b * frames)
    whole = bus.download().astype(np.float64)
    assert np.isfinite(whole).all() and np.abs(whole).max() > 1.0
    full.destroy()

    # two contiguous shards of the lane order, first 3 blocks
    acc = np.zeros((3 * frames, 2))
    sbus = gpu_ctx.bus(3 * frames)
    size = C.sizeof(T.WelshParams)
    for r in range(2):
        lo, hi = voice_range(V, r, 2)
        sp = (T.WelshParams * (hi - lo)).from_buffer_copy(bytes(memoryview(params))[lo * size:hi * size])
        shard = E.WelshSynth(gpu_ctx, sp)
        shard.handle_midi_events(T.note_events_np(np.arange(hi - lo, dtype=np.uint32), keys[lo:hi], True))
        for b in range(3):
            shard.render_mix(sbus, frames, at_frame=b * frames)
        acc += sbus.download().astype(np.float64)
        shard.destroy()
    assert np.max(np.abs(acc - whole[:3 * frames])) / V <= 1e-6, "sharded sum differs from the single-bank bus"

    # entity-boundary form: 5 voices per slot against the oracle
    mat = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(V, frames)
    bus2 = gpu_ctx.bus(blocks * frames)
    slot = voice % P.LIBRARY_SLOTS
    # (not key 57: 220 Hz over 44,100 Hz puts a waveform edge exactly on every 2,205th frame, a tie between the device's 64-bit phase
    # and the oracle's f64 one — docs/DSP_SPEC.md section 2)
    cand = np.array([0, 11, 500, 3000, 9000, 1, 12, 501, 3001, 9001, 2, 13])
    sample = []
    for s in range(P.LIBRARY_SLOTS):
        lanes_s = np.flatnonzero(slot == s)[cand]
        sample.extend(lanes_s[keys[lanes_s] != 57][:5].tolist())
    sample = np.sort(np.array(sample, dtype=np.int64))
    assert len(sample) == 5 * P.LIBRARY_SLOTS
    sub = (T.WelshParams * len(sample))()
    for k, i in enumerate(sample):
        sub[k] = params[int(i)]
    ob = oracle.Bank.welsh(sub)
    sl = np.arange(len(sample), dtype=np.uint32)
    row = np.empty(V, dtype=np.float32)
    worst = 0.0
    peak = np.ones(len(sample))
    for b in range(blocks):
        if b == 0:
            mat.handle_midi_events(on); ob.note_events(T.note_events_np(sl, keys[sample], True))
        if b == NOTE_OFF_BLOCK:
            mat.handle_midi_events(off); ob.note_events(T.note_events_np(sl, keys[sample], False))
        mat.generate_batch_values(block, frames)
        gpu_ctx.mix([block], frames, E._Slice(bus2, b * frames))
        want = ob.render(frames)
        peak = np.maximum(peak, np.abs(want).max(axis=(0, 1)))   # (a ripple of 10.7 rings above full scale: the bar is relative to it)
        dev = gpu_ctx.L.groove_block_device_ptr(block.h)
        for ch in (0, 1):
            for f in (0, 1, 63, 200, 255):
                lib.check(gpu_ctx.L.groove_download(gpu_ctx.h, C.c_void_p(dev + (ch * frames + f) * V * 4), row.ctypes.data_as(C.POINTER(C.c_float)), V), gpu_ctx.h)
                err = np.abs(row[sample].astype(np.float64) - want[ch, f, :]) / peak
                worst = max(worst, float(err.max()))
                assert err.max() <= 2e-5, f"block {b} ch {ch} frame {f}: {err.max():.3e} (slot {int(slot[sample[int(err.argmax())]])})"
    print(f"welsh-1m-library, {blocks} blocks, note-off at {NOTE_OFF_BLOCK}: worst sampled |gpu - oracle| / max(1, peak) = {worst:.3e}")
    mbus = bus2.download().astype(np.float64)
    per_block = np.abs(mbus - whole).reshape(blocks, -1).max(axis=1) / V
    assert per_block.max() <= 1e-6, per_block
    mat.destroy(); block.destroy(); bus.destroy(); bus2.destroy(); sbus.destroy()


def test_coefficient_look_ahead_is_the_per_lane_retune_bit_for_bit(gpu_ctx, oracle):
    """kernels.h "coefficient look-ahead": a wave whose live voices share the filter envelope's stage takes its filter coefficients from a
    table that lane j filled for frame k + j; a wave whose voices started apart retunes lane by lane.  The same voices through both: 256
    voices of ONE envelope-retuned patch in the per-kind (mix) kernels, (A) all struck in block 0 — every wave uniform — and (B) the odd
    voices struck a block later — every wave mixed.  The even voices, whose own history is the same in A and B, must come out the same
    BITS (a voice's samples do not depend on its neighbours), and both runs must match the oracle; for an f64-filter patch and for one
    the host lets filter in fp32.

    The LFO look-ahead (the smooth-f64 kinds: patch 2, a sine LFO on the pitch) is not the lanes' recurrences bit for bit — it evaluates
    exactly what they advance, 1e-16 a frame apart, which a phase accumulates — so with it on (the default) that patch's A and B need only
    agree to 2e-6 of full scale (on this patch they happen to agree to the bit as well: the two differ by less than the oscillators' fp32
    values resolve), and with it off (groove_set_look_ahead(1)) to the bit like the others.  (That the look-ahead runs is seen in the
    step time: tools/ab_env.sh, GROOVE_LOOK_AHEAD=1 against 3.)"""
    from groove_amd import entities as E
    old = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead
    gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = 0, 0, 1
    assert old[3] == 3
    try:
        n, frames, blocks = 256, 256, 14
        # benchmark patches: triangle LFO on the amplitude; sine LFO on the pitch (a smooth-f64 kind); no LFO, 49 Hz; sawtooth LFO on the
        # amplitude, 809 Hz — all four envelope-retuned; 8 and 24: a sine LFO on the cutoff (fp32 / f64 filter): coefficients from the
        # shared LFO's table.  The F32 kinds' LFO look-ahead evaluates the lanes' own expression on the same phase: the same bits.
        for j, look in ((1, 3), (2, 3), (2, 1), (3, 3), (13, 3), (8, 3), (24, 3), (24, 2)):
            gpu_ctx.look_ahead = look
            patch = P.welsh_patch(j)
            assert patch.filter_cutoff_end != 0.0 or j in (8, 24)
            params = (T.WelshParams * n)(*[patch] * n)
            keys = (40 + (np.arange(n) * 5) % 37).astype(np.uint8)
            keys[keys % 12 == 9] += 1
            lanes = np.arange(n, dtype=np.uint32)
            even, odd = lanes[::2], lanes[1::2]
            runs = {}
            for name, late in (("A", False), ("B", True)):
                synth, fused = E.WelshSynth(gpu_ctx, params), E.WelshSynth(gpu_ctx, params)   # (the fused kernels carry the fp32-filter bodies, whose table is fp32)
                assert "one launch per base kind" in synth.kernel_form(frames, False)
                block, bus = gpu_ctx.block(n, frames), gpu_ctx.bus(blocks * frames)
                ob = oracle.Bank.welsh(params)
                got, want = [], []
                for b in range(blocks):
                    evs = []
                    if b == 0:
                        evs.append(T.note_events_np(lanes if not late else even, keys if not late else keys[::2], True))
                    if b == 1 and late:
                        evs.append(T.note_events_np(odd, keys[1::2], True))
                    if b == 9:
                        evs.append(T.note_events_np(even, keys[::2], False))
                    if b == 11:
                        evs.append(T.note_events_np(odd, keys[1::2], False))
                    for ev in evs:
                        synth.handle_midi_events(ev); fused.handle_midi_events(ev); ob.note_events(ev)
                    synth.generate_batch_values(block, frames)
                    fused.render_mix(bus, frames, at_frame=b * frames)
                    got.append(block.download(frames)); want.append(ob.render(frames))
                got, want = np.concatenate(got, axis=1), np.concatenate(want, axis=1)
                per_voice = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
                assert per_voice.max() <= 1e-5, (j, name, per_voice.max())
                fbus = bus.download().astype(np.float64)
                assert np.sqrt(np.mean(((fbus - want.sum(axis=2).T) / n) ** 2)) <= 1e-6, (j, name)
                runs[name] = got
                synth.destroy(); fused.destroy(); block.destroy(); bus.destroy()
            a, b_ = runs["A"][:, :, ::2], runs["B"][:, :, ::2]
            assert np.abs(a).max() > 1e-2
            if j == 2 and look & 2:
                assert np.abs(a.astype(np.float64) - b_).max() <= 2e-6, f"patch {j}: exact LFO look-ahead against the lanes' recurrences: {np.abs(a.astype(np.float64) - b_).max():.3e}"
            else:
                assert np.array_equal(a.view(np.uint32), b_.view(np.uint32)), f"patch {j}: the even voices differ between the table and the per-lane retune"
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead = old


@pytest.mark.parametrize("kernel", ["mix", "all-kinds"])
def test_fast_bodies_are_the_shared_bodies_sample_for_sample(gpu_ctx, oracle, kernel):
    """kernels.h "FAST copies": in the fused mix kernel and the fused all-kinds kernel a wave whose live voices agree, when a block starts, on the filter envelope's
    record, the LFO's phase and the first-tick flag takes a copy of its body that holds the table frames' loop and nothing else.  Each of
    the 32 benchmark patches on 192 voices (three waves: one full, one of 61 voices struck a block later — uniform from then on —, one of 3),
    fused onto a bus three ways: look-ahead word 7 (FAST copies, counted), 1 (no LFO table: no wave is promised its tables, every wave
    takes the shared body) and 0.  The F32 kinds' buses must agree to the BIT (their tables hold the lanes' own arithmetic), the
    smooth-f64 kinds' to 2e-6 of full scale per voice (exact LFO evaluation against the recurrences); every bus matches the oracle; the
    FAST copies ran (fast_waves) and kept their promise (fast_table_misses)."""
    from groove_amd import entities as E
    old = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead
    gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves = 0, 0, (1 if kernel == "mix" else 1 << 20)
    try:
        n, frames, blocks = 192, 256, 10
        lanes = np.arange(n, dtype=np.uint32)
        keys = (40 + (lanes * 7) % 41).astype(np.uint8)
        keys[keys % 12 == 9] += 1
        a_set = np.r_[0:64, 128:131].astype(np.uint32)          # wave 0 (64 voices) and wave 2 (3 voices): struck in block 0
        b_set = np.arange(64, 125, dtype=np.uint32)             # wave 1: 61 voices, struck in block 2
        fast0 = gpu_ctx.debug_info()["fast_waves"]
        for j in range(P.N_PATCHES):
            patch = P.welsh_patch(j)
            params = (T.WelshParams * n)(*[patch] * n)
            ob = oracle.Bank.welsh(params)
            want = []
            buses = {}
            for look in (7, 1, 0):
                gpu_ctx.look_ahead = look
                synth = E.WelshSynth(gpu_ctx, params)
                assert ("mix_kernel" if kernel == "mix" else "any_kernel") in synth.kernel_form(frames, True), synth.kernel_form(frames, True)
                bus = gpu_ctx.bus(blocks * frames)
                for b in range(blocks):
                    evs = []
                    if b == 0:
                        evs.append(T.note_events_np(a_set, keys[a_set], True))
                    if b == 2:
                        evs.append(T.note_events_np(b_set, keys[b_set], True))
                    if b == 7:
                        evs.append(T.note_events_np(a_set, keys[a_set], False))
                    for ev in evs:
                        synth.handle_midi_events(ev)
                        if look == 7:
                            ob.note_events(ev)
                    synth.render_mix(bus, frames, at_frame=b * frames)
                    if look == 7:
                        want.append(ob.render_bus(frames))
                buses[look] = bus.download().astype(np.float64)
                synth.destroy(); bus.destroy()
            want = np.concatenate(want, axis=0)
            for look, got in buses.items():
                # (a bank of ONE patch: its voices' errors are the same error at other pitches and do not average out on the bus as a mixed
                # bank's do — the per-voice bar, 1e-5, is the one that applies; measured: 1.4e-6 at worst, patch 0)
                assert np.sqrt(np.mean(((got - want) / n) ** 2)) <= 5e-6, (j, look)
            smooth = bool(patch_is_smooth(patch))
            if smooth:
                assert np.abs(buses[7] - buses[1]).max() <= 2e-6 * n, (j, float(np.abs(buses[7] - buses[1]).max()))
            else:
                assert np.array_equal(buses[7], buses[1]), f"patch {j}: the FAST copy's bus differs from the shared body's"
            assert np.array_equal(buses[1], buses[0]), f"patch {j}: the coefficient look-ahead's bus differs from the per-lane retune's"
        info = gpu_ctx.debug_info()
        assert info["fast_waves"] - fast0 >= P.N_PATCHES * 3 * (blocks - 3) and info["fast_table_misses"] == 0, info
    finally:
        gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead = old


def patch_is_smooth(patch):
    """The patch routes its LFO to the pitch or the pulse width (the smooth-f64 or exact-f64 kinds: groove_lfo_routing 2, 3, 5, 6, 7)."""
    return int(patch.lfo_routing) in (T.LFO_PITCH, T.LFO_PULSE_WIDTH, T.LFO_PITCH_OSC2, T.LFO_PW_OSC1, T.LFO_PW_OSC2)
