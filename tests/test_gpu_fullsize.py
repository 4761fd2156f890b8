"""GPU tier, BASELINE.json's full size (1,000,000 Welsh voices): size-independent properties, plus a
sampled per-voice comparison with the oracle (which renders only the sampled voices).

  * sharding linearity: bus(all voices) == sum over 4 contiguous shards of bus(shard)  (SURVEY §8e);
  * silence in → silence out before any note-on; everything finite afterwards;
  * materialised voice block at full size: 512 sampled voices x sampled frames vs the f64 oracle;
  * fused and materialised forms agree at full size.
"""
import ctypes as C

import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T

pytestmark = pytest.mark.gpu
V = 1_000_000


def _subset(params, idx):
    out = (T.WelshParams * len(idx))()
    for k, i in enumerate(idx):
        out[k] = params[int(i)]
    return out


NOTE_OFF_BLOCK = 12
BLOCKS = 36   # note-on at block 0, note-off at block 12; the short-release patches (28, then 2 and 18) have BOTH envelopes
              # idle from blocks ~21 / ~33 on, so whole workgroups take the idle exit (kernels.h welsh_idle_workgroup)


def test_full_size_properties_and_sampled_parity(gpu_ctx, oracle):
    """1,000,000 voices through the kernels bench.py times (one kernel per base kind, class-specialised bodies, blocks
    pipelined): 36 blocks with the note-off inside, i.e. attack / decay / sustain, release, and the idle-workgroup exit."""
    from groove_amd import entities as E, lib
    params, vidx = P.welsh_voices_grouped(V)
    keys = (36 + (7 * vidx) % 49).astype(np.uint8)
    on, off = P.grouped_note_events(vidx, True), P.grouped_note_events(vidx, False)
    full = E.WelshSynth(gpu_ctx, params)
    frames, blocks = 256, BLOCKS
    bus = gpu_ctx.bus(blocks * frames)
    # silence before note-on
    full.render_mix(bus, frames)
    assert not bus.download(frames).any()
    full.destroy()

    # fused bus of the whole project
    full = E.WelshSynth(gpu_ctx, params)
    for b in range(blocks):
        if b == 0:
            full.handle_midi_events(on)
        if b == NOTE_OFF_BLOCK:
            full.handle_midi_events(off)
        full.render_mix(bus, frames, at_frame=b * frames)
    whole = bus.download().astype(np.float64)
    assert np.isfinite(whole).all() and np.abs(whole).max() > 1.0
    state = full.download_state()
    full.destroy()
    # the voices of the short-release patches really are idle at the end (so their workgroups took the idle exit)
    # WelshState words (dsp_core.h): 3 x OscState {u64 phase, x1, x2}, 2 x u64 increments, EnvState amp (16..22, stage first),
    # EnvState fil (23..29), 4 x f64 filter, vflags, pad; stage 0 = ENV_IDLE
    for patch in (28, 2, 18):
        lanes = np.nonzero(vidx % P.N_PATCHES == patch)[0]
        assert len(lanes) and (state[16, lanes] == 0).all() and (state[23, lanes] == 0).all(), f"patch {patch} is not idle at the end"
    assert (state[16] != 0).any(), "every voice idle: the long-release patches should still sound"

    # the same project as 4 contiguous shards, buses summed (what the multi-GPU path does), first 3 blocks
    from groove_amd.parallel import voice_range
    sb_blocks = 3
    acc = np.zeros((sb_blocks * frames, 2))
    sbus = gpu_ctx.bus(sb_blocks * frames)
    for r in range(4):
        lo, hi = voice_range(V, r, 4)
        sp = (T.WelshParams * (hi - lo)).from_buffer_copy(bytes(memoryview(params))[lo * C.sizeof(T.WelshParams):hi * C.sizeof(T.WelshParams)])
        shard = E.WelshSynth(gpu_ctx, sp)
        shard.handle_midi_events(T.note_events_np(np.arange(hi - lo, dtype=np.uint32), keys[lo:hi], True))
        for b in range(sb_blocks):
            shard.render_mix(sbus, frames, at_frame=b * frames)
        acc += sbus.download().astype(np.float64)
        shard.destroy()
    assert np.max(np.abs(acc - whole[:sb_blocks * frames])) / V <= 1e-6, "sharded sum differs from the single-bank bus"

    # materialised form at full size: sampled voices and frames against the oracle, at every block
    mat = E.WelshSynth(gpu_ctx, params)
    block = gpu_ctx.block(V, frames)
    bus2 = gpu_ctx.bus(blocks * frames)
    sample = np.unique((np.arange(512, dtype=np.int64) * 1953 + 7) % V)
    assert len(set((vidx[sample] % P.N_PATCHES).tolist())) == P.N_PATCHES  # every patch is in the sample
    ob = oracle.Bank.welsh(_subset(params, sample))
    sample_lanes = np.arange(len(sample), dtype=np.uint32)
    row = np.empty(V, dtype=np.float32)
    worst = 0.0
    for b in range(blocks):
        if b == 0:
            mat.handle_midi_events(on)
            ob.note_events(T.note_events_np(sample_lanes, keys[sample], True))
        if b == NOTE_OFF_BLOCK:
            mat.handle_midi_events(off)
            ob.note_events(T.note_events_np(sample_lanes, keys[sample], False))
        mat.generate_batch_values(block, frames)
        gpu_ctx.mix([block], frames, E._Slice(bus2, b * frames))  # reduces the render's own row sums
        want = ob.render(frames)
        dev = gpu_ctx.L.groove_block_device_ptr(block.h)
        for ch in (0, 1):
            for f in (0, 1, 63, 200, 255):
                off_b = (ch * frames + f) * V * 4
                lib.check(gpu_ctx.L.groove_download(gpu_ctx.h, C.c_void_p(dev + off_b), row.ctypes.data_as(C.POINTER(C.c_float)), V), gpu_ctx.h)
                err = row[sample].astype(np.float64) - want[ch, f, :]
                worst = max(worst, float(np.max(np.abs(err))))
                assert np.max(np.abs(err)) <= 2e-5, f"block {b} ch {ch} frame {f}: {np.max(np.abs(err)):.3e}"
    print(f"1,000,000 voices, {blocks} blocks, note-off at {NOTE_OFF_BLOCK}: worst sampled |gpu - oracle| = {worst:.3e}")
    mbus = bus2.download().astype(np.float64)
    assert np.max(np.abs(mbus - whole)) / V <= 1e-6, "fused and materialised buses differ"
    # ... block by block too: the release and the idle tail must agree, not only the loud part
    per_block = np.abs(mbus - whole).reshape(blocks, -1).max(axis=1) / V
    assert per_block.max() <= 1e-6, per_block
    mat.destroy(); block.destroy()


def test_repeated_renders_are_bit_identical(gpu_ctx):
    """The fused bus is a fixed-order reduction (partial rows, then segments), whichever streams the voice
    kernels ran on and however the blocks overlapped: two renders of the same project give the same bits.
    One size per launch form: all-kinds kernel (small bank), per-kind kernels with the block pipeline."""
    from groove_amd import entities as E
    for n in (40_000, 700_000):  # below and above the pipeline threshold (~550,000 voices, groove_hip.hip pipeline_min_waves)
        params, vidx = P.welsh_voices_grouped(n)
        on, off = P.grouped_note_events(vidx, True), P.grouped_note_events(vidx, False)
        runs = []
        for _ in range(2):
            synth = E.WelshSynth(gpu_ctx, params)
            bus = gpu_ctx.bus(12 * 256)
            for b in range(12):
                if b == 0:
                    synth.handle_midi_events(on)
                if b == 7:
                    synth.handle_midi_events(off)
                synth.render_mix(bus, 256, at_frame=b * 256)
            runs.append(bus.download())
            synth.destroy(); bus.destroy()
        assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32)), n


def test_million_voice_restarts_are_bit_identical_to_the_safe_stream_layout(gpu_ctx):
    """The window in which rounds 2-3 stalled: reset, note-on for every voice, the first blocks of
    1,000,000 voices through the per-kind pipelined kernels.  Three such restarts in the default stream layout, three in
    a second context with the safe layout (one priority, four streams): all six buses are the same bits, and neither
    context has counted a zero-frame segment (csrc/diag.h)."""
    import os
    from groove_amd import entities as E
    params, vidx = P.welsh_voices_grouped(V)
    on = P.grouped_note_events(vidx, True)
    frames, blocks = 256, 8

    def restarts(ctx):
        synth = E.WelshSynth(ctx, params)
        assert "blocks pipelined" in synth.kernel_form(frames, True)
        bus = ctx.bus(blocks * frames)
        out = []
        for _ in range(3):
            synth.reset()
            synth.handle_midi_events(on)
            for b in range(blocks):
                synth.render_mix(bus, frames, at_frame=b * frames)
            out.append(bus.download().view(np.uint32).copy())
        ctx.synchronize()
        zeros = ctx.debug_info()["zero_segments"]
        synth.destroy(); bus.destroy()
        return out, zeros

    a, za = restarts(gpu_ctx)
    old = os.environ.get("GROOVE_SAFE_STREAMS")
    os.environ["GROOVE_SAFE_STREAMS"] = "1"  # read by groove_init
    try:
        safe = E.Context(0)
    finally:
        if old is None:
            del os.environ["GROOVE_SAFE_STREAMS"]
        else:
            os.environ["GROOVE_SAFE_STREAMS"] = old
    assert safe.debug_info()["layout"].startswith("safe")
    b, zb = restarts(safe)
    safe.close()
    assert za == 0 and zb == 0, (za, zb)
    assert np.abs(a[0].view(np.float32)).max() > 1.0
    for k, run in enumerate(a + b):
        assert np.array_equal(run, a[0]), f"restart {k} differs"
