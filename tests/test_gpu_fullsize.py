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


def test_full_size_properties_and_sampled_parity(gpu_ctx, oracle):
    from groove_amd import entities as E, lib
    params, vidx = P.welsh_voices_grouped(V)
    keys = (36 + (7 * vidx) % 49).astype(np.uint8)
    on = P.grouped_note_events(vidx, True)
    full = E.WelshSynth(gpu_ctx, params)
    frames, blocks = 256, 3
    bus = gpu_ctx.bus(blocks * frames)
    # silence before note-on
    full.render_mix(bus, frames)
    assert not bus.download(frames).any()
    full.destroy()

    # fused bus of the whole project
    full = E.WelshSynth(gpu_ctx, params)
    full.handle_midi_events(on)
    for b in range(blocks):
        full.render_mix(bus, frames, at_frame=b * frames)
    whole = bus.download().astype(np.float64)
    assert np.isfinite(whole).all() and np.abs(whole).max() > 1.0
    full.destroy()

    # the same project as 4 contiguous shards, buses summed (what the multi-GPU path does)
    from groove_amd.parallel import voice_range
    acc = np.zeros_like(whole)
    sbus = gpu_ctx.bus(blocks * frames)
    for r in range(4):
        lo, hi = voice_range(V, r, 4)
        sp = (T.WelshParams * (hi - lo)).from_buffer_copy(bytes(memoryview(params))[lo * C.sizeof(T.WelshParams):hi * C.sizeof(T.WelshParams)])
        shard = E.WelshSynth(gpu_ctx, sp)
        shard.handle_midi_events(T.note_events_np(np.arange(hi - lo, dtype=np.uint32), keys[lo:hi], True))
        for b in range(blocks):
            shard.render_mix(sbus, frames, at_frame=b * frames)
        acc += sbus.download().astype(np.float64)
        shard.destroy()
    assert np.max(np.abs(acc - whole)) / V <= 1e-6, "sharded sum differs from the single-bank bus"

    # materialised form at full size: sampled voices and frames against the oracle
    mat = E.WelshSynth(gpu_ctx, params)
    mat.handle_midi_events(on)
    block = gpu_ctx.block(V, frames)
    bus2 = gpu_ctx.bus(blocks * frames)
    sample = np.unique((np.arange(512, dtype=np.int64) * 1953 + 7) % V)
    ob = oracle.Bank.welsh(_subset(params, sample))
    ob.note_events(T.note_events_np(np.arange(len(sample), dtype=np.uint32), keys[sample], True))
    row = np.empty(V, dtype=np.float32)
    dev = gpu_ctx.L.groove_block_device_ptr(block.h)
    for b in range(blocks):
        mat.generate_batch_values(block, frames)
        gpu_ctx.mix([block], frames, E._Slice(bus2, b * frames))
        want = ob.render(frames)
        for ch in (0, 1):
            for f in (0, 1, 63, 200, 255):
                off = (ch * frames + f) * V * 4
                lib.check(gpu_ctx.L.groove_download(gpu_ctx.h, C.c_void_p(dev + off), row.ctypes.data_as(C.POINTER(C.c_float)), V), gpu_ctx.h)
                err = row[sample].astype(np.float64) - want[ch, f, :]
                assert np.max(np.abs(err)) <= 2e-5, f"block {b} ch {ch} frame {f}: {np.max(np.abs(err)):.3e}"
    mbus = bus2.download().astype(np.float64)
    assert np.max(np.abs(mbus - whole)) / V <= 1e-6, "fused and materialised buses differ"
    mat.destroy(); block.destroy()


def test_repeated_renders_are_bit_identical(gpu_ctx):
    """The fused bus is a fixed-order reduction (partial rows, then segments), whichever streams the voice
    kernels ran on and however the blocks overlapped: two renders of the same project give the same bits.
    One size per launch form: all-kinds kernel (small bank), per-kind kernels with the block pipeline."""
    from groove_amd import entities as E
    for n in (40_000, 400_000):
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
