"""GPU: the role-split Welsh kernels (csrc/welsh_split.h — four wavefronts per 64 voices: envelopes + LFO / oscillators / cutoff tangent + coefficient quotients / filter +
gains, pipelined over the block's frames through LDS) computes every quantity with the serial kernels' statements in their
order: bus rows, voice blocks and the state record must be the serial kernels' BIT FOR BIT, for every patch of the
synthetic table (every waveform class, LFO routing, sync, both filter modes), through note-on, note-off, release, the idle
tail, ragged blocks, partly filled waves, and in the fused, block-writing and asynchronous forms.

(The serial kernels are run here WITHOUT their LFO look-ahead — groove_set_look_ahead(1): a serial wave whose voices share the LFO's phase
evaluates a pitch / pulse-width LFO exactly from the phase, lane = frame, where the role-split kernel's lanes advance the recurrences;
the two agree to 2e-6 — tests/test_gpu_random_inputs.py, tests/test_gpu_library.py — not to the bit.  The coefficient look-ahead stays on:
it is the lanes' own arithmetic.)"""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T

pytestmark = pytest.mark.gpu


@pytest.fixture()
def forms(gpu_ctx):
    """(set_form): switch the session ctx between the serial all-kinds kernel and the role-split one; restored afterwards."""
    old_tp, old_split, old_pipe, old_look = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead

    def set_form(split):
        gpu_ctx.time_parallel_max_voices = 0
        gpu_ctx.split_max_waves = 4096 if split else 0
        gpu_ctx.look_ahead = 1
    yield set_form
    gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves, gpu_ctx.pipeline_min_waves, gpu_ctx.look_ahead = old_tp, old_split, old_pipe, old_look


SIZES = [256, 256, 100, 256, 9, 255, 256, 8, 256, 64, 256, 256, 17, 256]


def _render(gpu_ctx, params, on, off, mode, sizes=SIZES, off_block=5):
    from groove_amd import entities as E
    n = len(params)
    synth = E.WelshSynth(gpu_ctx, params)
    total = sum(sizes)
    bus = gpu_ctx.bus(total)
    block = gpu_ctx.block(n, 256) if mode != "fused" else None
    blocks, at = [], 0
    for b, fr in enumerate(sizes):
        if b == 0:
            synth.handle_midi_events(on)
        if b == off_block:
            synth.handle_midi_events(off)
        if mode == "fused":
            synth.render_mix(bus, fr, at_frame=at)
        else:
            if mode == "async":
                synth.generate_batch_values_async(block, fr)
            else:
                synth.generate_batch_values(block, fr)
            gpu_ctx.mix([block], fr, E._Slice(bus, at))
            if b % 4 == 1:
                blocks.append(block.download(fr))
        at += fr
    form = synth.kernel_form(256, mode == "fused")
    state = synth.download_state()
    out = bus.download()
    synth.destroy(); bus.destroy()
    if block is not None:
        block.destroy()
    return out, blocks, state, form


@pytest.mark.parametrize("mode", ["fused", "block", "async"])
@pytest.mark.parametrize("n", [3072, 200])
def test_split_kernel_equals_the_serial_kernels_bit_for_bit(gpu_ctx, forms, mode, n):
    params, idx = P.welsh_voices_grouped(n, 0)      # every patch of the table; n = 200: partly filled waves and workgroups
    on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    forms(False)
    bus_s, blk_s, st_s, form_s = _render(gpu_ctx, params, on, off, mode)
    forms(True)
    bus_p, blk_p, st_p, form_p = _render(gpu_ctx, params, on, off, mode)
    assert "split" in form_p and "split" not in form_s, (form_s, form_p)
    assert np.abs(bus_s).max() > 1e-2
    assert np.array_equal(bus_s.view(np.uint32), bus_p.view(np.uint32)), "bus rows differ"
    for a, b in zip(blk_s, blk_p):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "voice blocks differ"
    assert np.array_equal(st_s, st_p), "state records differ"


def test_split_kernel_through_release_and_idle_tail(gpu_ctx, forms, oracle):
    """A longer render (the short-release patches go idle: whole workgroups take the idle exit) against the serial kernels
    bit for bit, and the bus against the oracle."""
    n = 4096
    params, idx = P.welsh_voices_grouped(n, 0)
    on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
    sizes = [256] * 40
    forms(False)
    bus_s, _, st_s, _ = _render(gpu_ctx, params, on, off, "fused", sizes, off_block=12)
    forms(True)
    bus_p, _, st_p, form = _render(gpu_ctx, params, on, off, "fused", sizes, off_block=12)
    assert "split" in form
    assert np.array_equal(bus_s.view(np.uint32), bus_p.view(np.uint32)) and np.array_equal(st_s, st_p)
    assert (st_p[16] == 0).any() and (st_p[16] != 0).any()   # some voices idle (amp envelope stage 0), some still sounding
    ob = oracle.Bank.welsh(params)
    want = []
    for b in range(len(sizes)):
        if b == 0:
            ob.note_events(on)
        if b == 12:
            ob.note_events(off)
        want.append(ob.render_bus(256))
    want = np.concatenate(want, axis=0) / n
    rms = float(np.sqrt(np.mean((bus_p.astype(np.float64) / n - want) ** 2)))
    assert rms <= 1e-5, rms


def test_split_is_the_default_for_mid_size_banks(gpu_ctx):
    """Above the time-parallel form's size and below the per-kind pipeline's: the role-split kernel, unless switched off."""
    from groove_amd import entities as E
    assert gpu_ctx.split_max_waves == 1024
    params, _ = P.welsh_voices_grouped(40_000, 0)
    s = E.WelshSynth(gpu_ctx, params)
    assert "split" in s.kernel_form(256, True) and "split" in s.kernel_form(256, False)
    assert "split" not in s.kernel_form(4, True)            # a handful of frames: nothing to pipeline
    s.destroy()
    mid = E.WelshSynth(gpu_ctx, P.welsh_voices_grouped(80_000, 0)[0])   # above one twelve-wave workgroup per CU: the all-kinds serial kernel since the end of
    assert "split" not in mid.kernel_form(256, True)                    # round 6 (its FAST bodies walk a block faster than the two-role form: groove_hip.hip split2_max_waves)
    mid.destroy()
    big = E.WelshSynth(gpu_ctx, P.welsh_voices_grouped(200_000, 0)[0])  # a second round of workgroups would cost more than the split saves
    assert "split" not in big.kernel_form(256, True)
    big.destroy()
    small = E.WelshSynth(gpu_ctx, P.welsh_voices_grouped(1024, 0)[0])
    assert "time-parallel" in small.kernel_form(256, True)
    small.destroy()


@pytest.mark.parametrize("roles", [2, 3, 4])
def test_every_role_count_equals_the_serial_kernels_bit_for_bit(oracle, roles):
    """The two-role form (front + tangent | back; two workgroups of eight wavefronts per CU: banks of up to 131,072 voices in
    one round), the three-role form (front | tangent | back) and the four-role form (envelopes + LFO | oscillators | tangent +
    coefficient quotients | back; sixteen wavefronts per CU), each selected with GROOVE_SPLIT_ROLES in a context of its own:
    bit-identical to the serial kernels, all three render forms."""
    import os
    from groove_amd import entities as E
    os.environ["GROOVE_SPLIT_ROLES"] = str(roles)
    word = {2: "two", 3: "three", 4: "four"}[roles] + " wavefronts"
    try:
        ctx = E.Context(0)
    finally:
        os.environ.pop("GROOVE_SPLIT_ROLES")
    try:
        ctx.look_ahead = 1   # (the module's docstring)
        for n in (3072, 200):
            params, idx = P.welsh_voices_grouped(n, 0)
            on, off = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
            for mode in ("fused", "block", "async"):
                ctx.time_parallel_max_voices = 0
                ctx.split_max_waves = 0
                bus_s, blk_s, st_s, form_s = _render(ctx, params, on, off, mode)
                ctx.split_max_waves = 4096
                bus_p, blk_p, st_p, form_p = _render(ctx, params, on, off, mode)
                assert word in form_p and "split" not in form_s, (form_s, form_p)
                assert np.abs(bus_s).max() > 1e-2
                assert np.array_equal(bus_s.view(np.uint32), bus_p.view(np.uint32)), (n, mode)
                for a, b in zip(blk_s, blk_p):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n, mode)
                assert np.array_equal(st_s, st_p), (n, mode)
    finally:
        ctx.close()
