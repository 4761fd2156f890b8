"""GPU tier: groove_bank_render_mix_deferred — the bus reduction of a lone time-parallel bank's block done by the bank's NEXT
render (welsh_tp.h tp_reduce_prev) or by whatever flushes it — against groove_bank_render_mix on an identical bank: Welsh, FM and
sampler banks, ragged block lengths, overwrite and accumulate, flushes by download / event record / explicit call in the middle of
a run, a bank too big for the form (falls back), and the project walk that uses it (config #2 against the oracle)."""
import os

import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T
from tests.seeds import drawn_seeds

pytestmark = pytest.mark.gpu


def _banks(gpu_ctx, kind, n):
    from groove_amd import entities as E
    from groove_amd import projects as PJ
    if kind == "welsh":
        params = P.welsh_voices(n)
        mk = lambda: E.WelshSynth(gpu_ctx, params)  # noqa: E731
        on, off = P.note_on_all(n), P.note_off_all(n)
    elif kind == "fm":
        spec = [s for s in PJ.plan("mixed-131072", np.arange(4 * n)) if s["kind"] == "fm"][0]
        mk = lambda: E.FmSynth(gpu_ctx, spec["params"])  # noqa: E731
        on, off = spec["events"][0], spec["events"][PJ.NOTE_OFF_BLOCK]
    else:
        spec = PJ.plan("sampler-16384", np.arange(n))[0]
        mk = lambda: E.Sampler(gpu_ctx, spec["pcm"], spec["descs"], spec["params"])  # noqa: E731
        ev = T.note_events_np(np.arange(n, dtype=np.uint32), np.full(n, 60, dtype=np.uint8), True)
        on, off = ev, None
    return mk, on, off


@pytest.mark.parametrize("kind,n", [("welsh", 256), ("welsh", 61), ("welsh", 700), ("welsh", 2048), ("welsh", 5000), ("fm", 200), ("sampler", 4096)])
def test_deferred_equals_immediate(gpu_ctx, kind, n):
    mk, on, off = _banks(gpu_ctx, kind, n)
    a = mk()
    frames_seq = [256, 256, 100, 256, 7, 1, 256, 255, 256, 256, 64, 256]
    total = sum(frames_seq)
    bus_a, bus_b = gpu_ctx.bus(total), gpu_ctx.bus(total)
    ev = gpu_ctx.event()
    # the deferred walk first (the form needs the bank to be the context's only one), then the immediate walk on a twin
    a.handle_midi_events(on)
    at = 0
    for i, fr in enumerate(frames_seq):
        if off is not None and i == 6:
            a.handle_midi_events(off)
        a.render_mix_deferred(bus_a, fr, accumulate=False, at_frame=at)
        if i == 3:
            gpu_ctx.flush_bus()
        if i == 5:
            gpu_ctx.record(ev)          # an event on the ctx stream: everything asked for so far is behind it
        if i == 8:
            part = bus_a.download()     # a download in the middle of the run
            assert np.isfinite(part).all()
        at += fr
    got = bus_a.download().astype(np.float64)
    # accumulate on top of the first pass: twice the bus
    a.reset(); a.handle_midi_events(on)
    at = 0
    for i, fr in enumerate(frames_seq):
        if off is not None and i == 6:
            a.handle_midi_events(off)
        a.render_mix_deferred(bus_a, fr, accumulate=True, at_frame=at)
        at += fr
    got2 = bus_a.download().astype(np.float64)
    a.destroy()
    b = mk()
    b.handle_midi_events(on)
    at = 0
    for i, fr in enumerate(frames_seq):
        if off is not None and i == 6:
            b.handle_midi_events(off)
        b.render_mix(bus_b, fr, accumulate=False, at_frame=at)
        at += fr
    want = bus_b.download().astype(np.float64)
    b.destroy(); bus_a.destroy(); bus_b.destroy()
    scale = max(1e-3, float(np.abs(want).max()))
    assert float(np.sqrt(np.mean(want ** 2))) > 1e-4
    assert np.abs(got - want).max() <= 2e-6 * scale * max(1.0, np.sqrt(n) / 8)
    assert np.abs(got2 - 2.0 * want).max() <= 4e-6 * scale * max(1.0, np.sqrt(n) / 8)


def test_deferred_falls_back_for_banks_it_does_not_fit(gpu_ctx):
    """More than 2,048 partial rows (8,196 Welsh voices in the one-voice form: 2,049 workgroups): groove_bank_render_mix itself,
    bit for bit."""
    from groove_amd import entities as E
    n = 8196
    params = P.welsh_voices(n)
    on = P.note_on_all(n)
    buses = []
    for deferred in (True, False):
        s = E.WelshSynth(gpu_ctx, params)
        s.handle_midi_events(on)
        bus = gpu_ctx.bus(4 * 256)
        for b in range(4):
            (s.render_mix_deferred if deferred else s.render_mix)(bus, 256, accumulate=False, at_frame=b * 256)
        buses.append(bus.download())
        s.destroy(); bus.destroy()
    assert np.array_equal(buses[0], buses[1])


def test_config2_project_walk_uses_the_deferred_form(gpu_ctx, oracle):
    from groove_amd import projects as PJ
    from oracle.projects import OracleProject
    sel = np.arange(256)
    proj = PJ.Project(gpu_ctx, "welsh-256", sel)
    bus = gpu_ctx.bus(60 * 256)
    for b in range(60):
        proj.step(bus, b * 256)
    got = bus.download().astype(np.float64) / 256
    want = OracleProject("welsh-256", sel).render(60) / 256
    proj.destroy(); bus.destroy()
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6


def test_banks_taking_turns_equal_banks_side_by_side(gpu_ctx, oracle):
    """Config #5's per-GPU share (Welsh + FM + sampler banks, 2,048 voices here): the banks in turn on the ctx stream, each
    render carrying the reduction of the one before it, against the side-by-side form (GROOVE_TAKE_TURNS_MAX_VOICES=0) and the
    oracle."""
    from groove_amd import projects as PJ
    from oracle.projects import OracleProject
    sel = np.arange(2048)
    buses = []
    old = PJ.TAKE_TURNS_MAX_VOICES
    try:
        for limit in (16384, 0):
            PJ.TAKE_TURNS_MAX_VOICES = limit
            proj = PJ.Project(gpu_ctx, "mixed-131072", sel)
            assert proj.take_turns == (limit > 0)
            bus = gpu_ctx.bus(40 * 256)
            for b in range(40):
                proj.step(bus, b * 256)
            buses.append(bus.download().astype(np.float64) / len(sel))
            proj.destroy(); bus.destroy()
    finally:
        PJ.TAKE_TURNS_MAX_VOICES = old
    want = OracleProject("mixed-131072", sel).render(40) / len(sel)
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    for got in buses:
        assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6
    assert np.abs(buses[0] - buses[1]).max() <= 1e-6


@pytest.mark.parametrize("voices,blocks", [(16384, 100), (2048, 40), (600, 40), (30000, 20)])
def test_small_mixed_project_in_one_launch(gpu_ctx, oracle, monkeypatch, voices, blocks):
    """groove_banks_render_mix_deferred (welsh_tp.h tp_mixed_kernel): config #5's per-GPU share — 8,192 Welsh (two per wavefront)
    + 4,096 FM (four per wavefront) + 4,096 sampler voices — and smaller mixes (one Welsh voice per wavefront; one FM voice per
    wavefront) in ONE launch per block, against the banks in turn (each render carrying the reduction of the one before it), the
    oracle on a voice sample of every kind, and itself (bit-reproducible from a reset state).  The 100-block run passes the
    project's note-off (block 86) and the sampler's staggered note-ons ride in the kernel arguments every block.  A project the
    form does not take (30,000 voices: more than 2,048 rows) goes bank by bank: the same bits as the banks in turn."""
    from groove_amd import projects as PJ
    from oracle.projects import OracleProject
    sel = np.arange(voices, dtype=np.int64)
    buses = {}
    monkeypatch.setattr(PJ, "TAKE_TURNS_MAX_VOICES", max(PJ.TAKE_TURNS_MAX_VOICES, voices))
    for one in (True, False):
        proj = PJ.Project(gpu_ctx, "mixed-131072", sel, one_launch=one)
        assert proj.take_turns and proj.one_launch == one
        runs = []
        for rep in range(2 if one else 1):
            if rep:
                proj.reset()
            bus = gpu_ctx.bus(blocks * 256)
            for b in range(blocks):
                proj.step(bus, b * 256)
            runs.append(bus.download().astype(np.float64) / voices)
            bus.destroy()
        proj.destroy()
        if one:
            assert np.array_equal(runs[0], runs[1])
        buses[one] = runs[0]
    assert np.sqrt(np.mean(buses[False] ** 2)) > 1e-3
    if voices > 16384:
        assert np.array_equal(buses[True], buses[False])
    else:
        assert np.abs(buses[True] - buses[False]).max() <= 1e-6
    # a sample of every kind through the SAME form (the project's voice rule depends on the index alone), against the oracle
    samp = np.unique(np.concatenate([np.arange(0, min(voices, 400)), np.arange(voices - min(voices, 200), voices)])).astype(np.int64)
    proj = PJ.Project(gpu_ctx, "mixed-131072", samp, one_launch=True)
    nb = min(blocks, 100)
    bus = gpu_ctx.bus(nb * 256)
    for b in range(nb):
        proj.step(bus, b * 256)
    got = bus.download().astype(np.float64) / len(samp)
    proj.destroy(); bus.destroy()
    want = OracleProject("mixed-131072", samp).render(nb) / len(samp)
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6


@pytest.mark.parametrize("n,blocks", [(20000, 100), (70000, 30), (140000, 30)])
def test_deferred_serial_small_bank_forms(gpu_ctx, n, blocks):
    """Welsh banks too big for the time-parallel form and too small for the per-kind pipeline (the role-split kernels: 20,000
    and 70,000 voices; the all-kinds kernel: 140,000): the deferred form — the next block's kernel sums the previous block's
    rows, the workgroups of released voices included (they do their share before they leave: the 100-block run passes the
    project's note-off at block 86) — against groove_bank_render_mix."""
    from groove_amd import projects as PJ
    buses, forms = [], []
    for deferred in (True, False):
        proj = PJ.Project(gpu_ctx, "welsh-1m", np.arange(n, dtype=np.int64))
        forms.append(proj.banks[0][0].kernel_form(256, True))
        proj.take_turns = deferred
        bus = gpu_ctx.bus(blocks * 256)
        for b in range(blocks):
            proj.step(bus, b * 256)
        buses.append(bus.download().astype(np.float64) / n)
        proj.destroy(); bus.destroy()
    assert "role-split" in forms[0] or "all base kinds" in forms[0], forms
    assert np.sqrt(np.mean(buses[1] ** 2)) > 1e-3
    assert np.abs(buses[0] - buses[1]).max() <= 2e-6


def test_mix_deferred_rides_in_the_next_chain_launch_or_a_flush(gpu_ctx):
    """groove_mix_deferred: a chain's lane sums are put on the bus by the NEXT effect-chain launch on the ctx stream, or by
    groove_bus_flush / anything that waits for the ctx stream — the same bus as groove_mix (1,024-lane groups: <= 8 rows, equal
    to fp32 rounding; here 2 rows: the same bits).  A block without valid lane sums is mixed at once."""
    from groove_amd import entities as E
    n, frames, blocks = 2048, 256, 6
    params, vidx = P.welsh_voices_grouped(n)
    on = P.grouped_note_events(vidx, True)
    fxp = lambda **kw: (T.FxParams * n)(*[T.fx_params(**kw) for _ in range(n)])
    outs = []
    for deferred in (False, True):
        synth = E.WelshSynth(gpu_ctx, params)
        chain = [E.Effect(gpu_ctx, T.FX_GAIN, fxp(ceiling=0.5)), E.Effect(gpu_ctx, T.FX_DELAY, fxp(delay_seconds=0.01))]  # one fused run launch
        block = gpu_ctx.block(n, frames)
        bus = gpu_ctx.bus(blocks * frames)
        synth.handle_midi_events(on)
        for b in range(blocks):
            synth.generate_batch_values(block, frames)
            gpu_ctx.transform_chain(chain, block, frames)      # in the deferred walk this launch carries block b - 1's rows
            if deferred:
                gpu_ctx.mix_deferred(block, frames, E._Slice(bus, b * frames))
            else:
                gpu_ctx.mix([block], frames, E._Slice(bus, b * frames))
        outs.append(bus.download())                            # (a download waits for the ctx stream: the last block is flushed)
        # a block whose lane sums are gone (an upload) takes the plain path at once
        block.upload(np.ones((2, frames, n), dtype=np.float32))
        bus2 = gpu_ctx.bus(frames)
        gpu_ctx.mix_deferred(block, frames, bus2)
        assert np.allclose(bus2.download(), float(n), rtol=1e-6)
        for e in chain:
            e.destroy()
        synth.destroy(); block.destroy(); bus.destroy(); bus2.destroy()
    assert np.abs(outs[0]).max() > 1.0
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_paced_render_mix_gives_the_bus_of_render_mix(gpu_ctx):
    """groove_bank_render_mix_paced — several banks side by side, the host waiting for the events, every bank's bus reduction
    launched by its NEXT paced call or a flush point — against groove_bank_render_mix on identical banks: the same bits (the
    reduction kernels and their order on the bus are the same), with note events in the middle, ragged blocks, a flush by a
    download half way, and an unpaced call in between."""
    from groove_amd import entities as E, projects as PJ
    sel = np.arange(6000, dtype=np.int64)
    lens = [256, 256, 100, 256, 33, 256, 256, 256]
    outs = []
    for paced in (False, True):
        specs = PJ.plan("mixed-131072", sel)
        banks = []
        for spec in specs:
            if spec["kind"] == "welsh":
                inst = E.WelshSynth(gpu_ctx, spec["params"])
            elif spec["kind"] == "fm":
                inst = E.FmSynth(gpu_ctx, spec["params"])
            else:
                inst = E.Sampler(gpu_ctx, spec["pcm"], spec["descs"], spec["params"])
            banks.append((inst, spec["events"]))
        bus = gpu_ctx.bus(sum(lens))
        at, mid = 0, None
        for b, frames in enumerate(lens):
            for i, (inst, events) in enumerate(banks):
                ev = events.get(b)
                if ev is not None:
                    inst.handle_midi_events(ev)
                if paced and not (b == 5 and i == 1):     # (one unpaced call in the middle: it flushes what is pending first)
                    inst.render_mix_paced(bus, frames, accumulate=i > 0, at_frame=at)
                else:
                    inst.render_mix(bus, frames, accumulate=i > 0, at_frame=at)
            at += frames
            if b == 3:
                mid = bus.download(at).copy()             # a flush point: every block so far is complete
        outs.append((mid, bus.download()))
        for inst, _ in banks:
            inst.destroy()
        bus.destroy()
    assert np.abs(outs[0][1]).max() > 1.0
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


def _mixed_banks(ctx, sel, oracle=None):
    """The instruments of a mixed project (groove_amd.projects.plan); with `oracle`, the same banks on the oracle as a third element."""
    from groove_amd import entities as E, projects as PJ
    banks = []
    for spec in PJ.plan("mixed-131072", sel):
        if spec["kind"] == "welsh":
            inst = E.WelshSynth(ctx, spec["params"])
            ob = oracle.Bank.welsh(spec["params"]) if oracle else None
        elif spec["kind"] == "fm":
            inst = E.FmSynth(ctx, spec["params"])
            ob = oracle.Bank.fm(spec["params"]) if oracle else None
        else:
            inst = E.Sampler(ctx, spec["pcm"], spec["descs"], spec["params"])
            ob = oracle.Bank.sampler(spec["pcm"], spec["descs"], spec["params"]) if oracle else None
        banks.append((inst, spec["events"], ob) if oracle else (inst, spec["events"]))
    return banks


def test_a_deferred_render_flushes_pending_paced_reductions_first(gpu_ctx):
    """include/groove_hip.h: any unpaced render or mix puts the pending PACED reductions on their buses first.  A paced bank that
    OVERWRITES the bus followed by a deferred bank that ACCUMULATES onto it (round 4's advisor: the deferred call skipped the
    flush, bus_flush later applied the deferred rows first and the paced, non-accumulating reduction then overwrote them)."""
    sel = np.arange(3000, dtype=np.int64)
    outs = []
    for mode in ("plain", "paced+deferred"):
        banks = _mixed_banks(gpu_ctx, sel)
        bus = gpu_ctx.bus(5 * 256)
        for b in range(5):
            for i, (inst, events) in enumerate(banks):
                if events.get(b) is not None:
                    inst.handle_midi_events(events[b])
                if mode == "plain":
                    inst.render_mix(bus, 256, accumulate=i > 0, at_frame=b * 256)
                elif i == 0:
                    inst.render_mix_paced(bus, 256, accumulate=False, at_frame=b * 256)
                else:
                    inst.render_mix_deferred(bus, 256, accumulate=True, at_frame=b * 256)
        outs.append(bus.download().astype(np.float64))
        for inst, _ in banks:
            inst.destroy()
        bus.destroy()
    scale = float(np.abs(outs[0]).max())
    assert scale > 1.0
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6 * scale * np.sqrt(len(sel)) / 8


@pytest.mark.parametrize("safe_streams", [False, True])
def test_a_paced_call_whose_wait_times_out_loses_no_block(safe_streams, monkeypatch):
    """groove_bank_render_mix_paced's host wait for the previous block's render passes its deadline (a side stream blocked on
    purpose): the call reports it — and NOTHING is forgotten: the previous block's reduction is queued behind device-side waits,
    the new block is registered, and a caller that carries on gets the bus of an undisturbed run, bit for bit.  Under
    GROOVE_SAFE_STREAMS=1 (the bank streams are the kind streams: a blocked stream holds back more) the deadline that passes is
    the OTHER host wait of the call, for the reduction that frees the slot's rows: until the end of round 5 that one returned before
    the render, and the caller that carried on had skipped a block."""
    from groove_amd import entities as E, lib
    if safe_streams:
        monkeypatch.setenv("GROOVE_SAFE_STREAMS", "1")
    sel = np.arange(3000, dtype=np.int64)
    outs = []
    for disturbed in (False, True):
        ctx = E.Context(0)
        try:
            banks = _mixed_banks(ctx, sel)
            bus = ctx.bus(6 * 256)
            timeouts = 0
            for b in range(6):
                if disturbed and b == 2:
                    for k in range(16):
                        try:
                            ctx.debug_spin(k, 700)      # every side stream busy for 0.7 s: block 2's renders queue behind it
                        except lib.GrooveError:
                            break
                for i, (inst, events) in enumerate(banks):
                    if events.get(b) is not None:
                        inst.handle_midi_events(events[b])
                    if disturbed and b == 3:
                        ctx.sync_timeout_ms = 100        # the wait for block 2's render cannot make it
                    try:
                        inst.render_mix_paced(bus, 256, accumulate=i > 0, at_frame=b * 256)
                    except lib.GrooveError as e:
                        assert "not complete after 100 ms" in str(e), str(e)
                        timeouts += 1
                    ctx.sync_timeout_ms = 20000
            outs.append(bus.download().copy())
            assert (1 <= timeouts <= len(banks)) if disturbed else timeouts == 0, timeouts
            assert ctx.debug_info()["zero_segments"] == 0
        finally:
            ctx.close()
    assert np.abs(outs[0]).max() > 1.0
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_mix_deferred_refuses_a_block_of_another_ctx(gpu_ctx):
    from groove_amd import entities as E, lib
    other = E.Context(0)
    try:
        n = 128
        synth = E.WelshSynth(other, P.welsh_voices(n))
        synth.handle_midi_events(P.note_on_all(n))
        block = other.block(n, 256)
        synth.generate_batch_values(block, 256)
        bus = gpu_ctx.bus(256)
        with pytest.raises(lib.GrooveError, match="another ctx"):
            gpu_ctx.mix_deferred(block, 256, bus)
        bus.destroy()
    finally:
        other.close()


def test_deferred_forms_are_bit_reproducible_run_to_run(gpu_ctx):
    """The forms bench.py times for configs #2, #4 and #5's per-GPU share (groove_bank_render_mix_deferred: a lone bank, and banks
    of different row counts taking turns): two identical call sequences from a reset state give identical bits — the FIRST run
    included (both row buffers are sized for the largest bank before anything is pending, so no block of the first run is flushed
    through the other reduction order).  What is NOT promised: equal bits across different call patterns (include/groove_hip.h)."""
    from groove_amd import entities as E, projects as PJ
    for workload, sel, blocks in (("welsh-256", np.arange(256), 40), ("sampler-16384", np.arange(16384), 60), ("mixed-131072", np.arange(16384), 40),
                                  ("chain-4096", np.arange(4096), 72)):   # (config #3: the paced render-ahead walk with groove_mix_deferred; its all-wet chorus and
                                                                           #  delay lines keep the bus silent for the first 15,435 frames)
        ctx = E.Context(0)   # a fresh ctx: its first run is the project's first run
        try:
            proj = PJ.Project(ctx, workload, sel.astype(np.int64))
            assert proj.take_turns or (proj.paced and proj.ahead_walk), workload
            runs = []
            for rep in range(3):
                if rep:
                    proj.reset()
                bus = ctx.bus(blocks * 256)
                for b in range(blocks):
                    proj.step(bus, b * 256)
                runs.append(bus.download().copy())
                bus.destroy()
            proj.destroy()
        finally:
            ctx.close()
        assert np.abs(runs[0]).max() > 0.1, workload
        assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32)), workload
        assert np.array_equal(runs[1].view(np.uint32), runs[2].view(np.uint32)), workload


def _chain_bits(ctx, ap, frames_of, probe=(), n=512, between=None, destroy_first=False, cap=256):
    """A 512-voice Welsh bank through BiQuad -> Delay -> Reverb, render-ahead by hand (the calls bench.py's paced walk makes) with the
    all-pass stream on or off; returns the bus and the probed blocks' content."""
    from groove_amd import entities as E, patches as P
    assert not ctx.fx_allpass_stream
    ctx.fx_allpass_stream = ap
    params = P.welsh_voices(n)
    synth = E.WelshSynth(ctx, params)
    fxp = (T.FxParams * n)(*[T.fx_params(cutoff_hz=900.0 + 13 * (i % 40), delay_seconds=0.05, attenuation=0.9, reverb_seconds=0.8) for i in range(n)])
    fx = [E.Effect(ctx, T.FX_BIQUAD_LP12, fxp), E.Effect(ctx, T.FX_DELAY, fxp), E.Effect(ctx, T.FX_REVERB, fxp)]
    rot = [ctx.block(n, cap) for _ in range(4)]
    total = sum(frames_of)
    bus = ctx.bus(total)
    synth.handle_midi_events(P.note_on_all(n))
    got = {}
    at = 0
    for b, fr in enumerate(frames_of):
        blk = rot[b % 4]
        blk.wait_released()
        synth.generate_batch_values_async(blk, fr)
        blk.wait_ready()
        ctx.transform_chain(fx, blk, fr)
        ctx.mix_deferred(blk, fr, E._Slice(bus, at), accumulate=False)
        if b in probe:
            got[b] = blk.download(fr).copy()       # a reader of the block: ordered behind its all-passes, wherever they run
        if between is not None:
            between(b, fx, bus, at)
        blk.release()
        at += fr
    if destroy_first:   # the last block's lane sums still wait for an all-pass launch that will not come: the block's destruction flushes them
        for blk in rot:
            blk.destroy()
    out = bus.download().copy()
    for e in fx:
        e.destroy()
    if not destroy_first:
        for blk in rot:
            blk.destroy()
    synth.destroy(); bus.destroy()
    ctx.fx_allpass_stream = False
    return out, got


def test_allpass_stream_gives_the_same_bits(gpu_ctx):
    """groove_set_fx_allpass_stream: a chain's closing reverb leaves its all-passes — the block's last kernel — on a side stream, the
    lane sums reach the bus through the NEXT all-pass launch.  Same kernels, same arithmetic, same order of sums: the bus and the
    blocks are bit-identical to the ctx-stream form, on whole and on ragged blocks (a 100-frame block takes the same path, a
    2,000-frame one the chunked all-pass kernel on the ctx stream: the hand-over between the two forms is ordered)."""
    shapes = ([256] * 40, [256] * 6 + [100, 256, 37, 256, 256, 1, 256] + [256] * 6, [256, 256, 2000, 256, 300, 1024, 256, 2048, 100, 256, 256, 512, 256])
    for frames_of in shapes:
        cap = max(frames_of)
        a, ga = _chain_bits(gpu_ctx, False, frames_of, probe=(0, 5, 9, len(frames_of) - 1), cap=cap)
        b, gb = _chain_bits(gpu_ctx, True, frames_of, probe=(0, 5, 9, len(frames_of) - 1), cap=cap)
        assert np.abs(a).max() > 1e-3
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for k in ga:
            assert np.array_equal(ga[k].view(np.uint32), gb[k].view(np.uint32)), k
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_allpass_stream_is_left_to_the_all_passes(gpu_ctx):
    """Banks take the library's three bank streams in turn; the one the all-passes use is skipped while the knob is on, and a bank that
    sat on it moves (bench.py's config #3 as the fourth project of a run rendered on that very stream: 0.065 ms per block instead of
    0.040).  Whatever stream the bank lands on, the bits are the same."""
    from groove_amd import entities as E, patches as P
    ref, _ = _chain_bits(gpu_ctx, False, [256] * 12)
    spare = []
    for shift in range(3):                      # the next bank's slot moves on by one each time
        spare.append(E.WelshSynth(gpu_ctx, P.welsh_voices(64)))
        got, _ = _chain_bits(gpu_ctx, True, [256] * 12)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), shift
    for s in spare:
        s.destroy()


def test_allpass_stream_flush_points_keep_every_block(gpu_ctx):
    """Whatever interrupts the walk — a bus download in the middle (a flush: the ctx stream sums the pending rows itself), a parameter
    change of the reverb, a reset of the chain — the bus has every block and the same bits as the ctx-stream form."""
    def meddle(b, fx, bus, at):
        if b == 7:
            bus.download()                                        # flush point: rows pending on the all-pass stream
        if b == 11:
            fx[2].control_set_param_by_index(T.CTL_FX_ATTENUATION, 0.7)   # the reverb's parameters: its lines are settled first
        if b == 15:
            fx[0].control_set_param_by_index(T.CTL_FX_CUTOFF, 0.6)
    a, _ = _chain_bits(gpu_ctx, False, [256] * 24, between=meddle)
    b, _ = _chain_bits(gpu_ctx, True, [256] * 24, between=meddle)
    assert np.abs(a[20 * 256:]).max() > 1e-3
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c, _ = _chain_bits(gpu_ctx, True, [256] * 24, destroy_first=True)     # the blocks die before anybody has asked for the bus
    d, _ = _chain_bits(gpu_ctx, False, [256] * 24)
    assert np.abs(c[23 * 256:]).max() > 1e-3 and np.array_equal(c.view(np.uint32), d.view(np.uint32))


def test_allpass_stream_keeps_the_order_of_a_bus_with_several_sources(gpu_ctx):
    """Two chained instruments and a small fused bank onto ONE bus slice per block (accumulate = 0, 1, 1): the deferred lane sums of
    the chains wait on the all-pass stream, the bank's on the ctx stream — whichever is pending when the other kind arrives is flushed
    first, so the bus is the ctx-stream form's bit for bit."""
    from groove_amd import entities as E, patches as P

    def run(ap):
        assert not gpu_ctx.fx_allpass_stream
        gpu_ctx.fx_allpass_stream = ap
        n = 256
        synths = [E.WelshSynth(gpu_ctx, P.welsh_voices(n)) for _ in range(2)]
        lone = E.FmSynth(gpu_ctx, P.fm_voices(128))
        chains = []
        for c in range(2):
            fxp = (T.FxParams * n)(*[T.fx_params(cutoff_hz=700.0 + 300 * c + 11 * (i % 30), delay_seconds=0.03 + 0.02 * c, attenuation=0.85, reverb_seconds=0.6 + 0.3 * c) for i in range(n)])
            chains.append([E.Effect(gpu_ctx, T.FX_DELAY, fxp), E.Effect(gpu_ctx, T.FX_REVERB, fxp)])
        rots = [[gpu_ctx.block(n, 256) for _ in range(4)] for _ in range(2)]
        blocks = 20
        bus = gpu_ctx.bus(blocks * 256)
        for s in synths:
            s.handle_midi_events(P.note_on_all(n))
        lone.handle_midi_events(P.note_on_all(128))
        for b in range(blocks):
            at = E._Slice(bus, b * 256)
            for c in range(2):
                blk = rots[c][b % 4]
                blk.wait_released()
                synths[c].generate_batch_values_async(blk, 256)
                blk.wait_ready()
                gpu_ctx.transform_chain(chains[c], blk, 256)
                gpu_ctx.mix_deferred(blk, 256, at, accumulate=c > 0)
                blk.release()
            lone.render_mix_deferred(bus, 256, accumulate=True, at_frame=b * 256)
        out = bus.download().copy()
        for c in range(2):
            for e in chains[c]:
                e.destroy()
            for blk in rots[c]:
                blk.destroy()
            synths[c].destroy()
        lone.destroy(); bus.destroy()
        gpu_ctx.fx_allpass_stream = False
        return out

    a, b = run(False), run(True)
    assert np.abs(a).max() > 1e-2
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_allpass_stream_random_walks(gpu_ctx, oracle):
    """Seeded random call sequences over two chained instruments (one chain ends in a reverb, the other in a delay — whose run is its
    last kernel and stays on the ctx stream): per block each instrument is rendered ahead or in place, its chain run whole or stage
    by stage, mixed deferred / at once / not at all, released or not, waited for or not, with downloads, parameter changes, ragged
    blocks and effect resets in between.  Every sequence is played twice, all-pass stream off and on: same buses, same blocks; and a
    third time as the plain walk — rendered in place, stage by stage, mixed at once — whose bus the others must match to rounding and
    which itself is held to the f64 oracle playing the same sequence."""
    from groove_amd import entities as E, patches as P

    def play(seed, ap, plain=False):
        rng = np.random.default_rng(seed)
        assert not gpu_ctx.fx_allpass_stream
        gpu_ctx.fx_allpass_stream = ap
        n = 192
        synths = [E.WelshSynth(gpu_ctx, P.welsh_voices(n, first_voice=7 * c)) for c in range(2)]
        fxp = [(T.FxParams * n)(*[T.fx_params(cutoff_hz=600.0 + 400 * c + 9 * (i % 50), delay_seconds=0.02 + 0.015 * c, attenuation=0.8, reverb_seconds=0.5 + 0.4 * c) for i in range(n)]) for c in range(2)]
        chains = [[E.Effect(gpu_ctx, T.FX_BIQUAD_LP12, fxp[0]), E.Effect(gpu_ctx, T.FX_DELAY, fxp[0]), E.Effect(gpu_ctx, T.FX_REVERB, fxp[0])],
                  [E.Effect(gpu_ctx, T.FX_REVERB, fxp[1]), E.Effect(gpu_ctx, T.FX_DELAY, fxp[1])]]
        rots = [[gpu_ctx.block(n, 256) for _ in range(3)] for _ in range(2)]
        blocks = 40   # (~8,000 frames: the reverb's combs are 1,100 - 1,600 frames long, the walk has to outlast them)
        frames_of = [int(rng.choice([256, 256, 256, 256, 256, 100, 37, 1])) for _ in range(blocks)]
        bus = gpu_ctx.bus(sum(frames_of))
        seen = []
        kinds = [[T.FX_BIQUAD_LP12, T.FX_DELAY, T.FX_REVERB], [T.FX_REVERB, T.FX_DELAY]]
        if plain:          # the oracle beside the plain walk: the same voices, chains, parameter changes and resets in f64
            ob = [oracle.Bank.welsh(P.welsh_voices(n, first_voice=7 * c)) for c in range(2)]
            cur = [[fxp[c] for _ in kinds[c]] for c in range(2)]
            ofx = [[oracle.Fx(k, cur[c][i]) for i, k in enumerate(kinds[c])] for c in range(2)]
            for o_ in ob:
                o_.note_events(P.note_on_all(n))
            want = np.zeros((sum(frames_of), 2))
        for s in synths:
            s.handle_midi_events(P.note_on_all(n))
        at = 0
        for b, fr in enumerate(frames_of):
            first = True
            for c in range(2):
                blk = rots[c][b % 3]
                if rng.random() < 0.6:
                    blk.wait_released()
                done, r = 0, rng.random()
                r2, r3, r4 = rng.random(), rng.random(), rng.random()   # (drawn whatever the branch: the plain walk stays aligned)
                if plain:          # the reference walk: rendered in place, stage by stage, mixed at once
                    synths[c].generate_batch_values(blk, fr)
                elif r < 0.25:     # the render with the chain's leading IIR stage behind it on its side stream (fused into the kernel where it can be)
                    done = synths[c].generate_batch_values_chain_async(blk, chains[c], fr)
                elif r < 0.7:
                    synths[c].generate_batch_values_async(blk, fr)
                    if r2 < 0.3:
                        done = gpu_ctx.transform_chain_async(chains[c], blk, fr)
                else:
                    synths[c].generate_batch_values(blk, fr)
                if not plain and r < 0.7 and r3 < 0.6:
                    blk.wait_ready()
                if not plain and r4 < 0.75:
                    gpu_ctx.transform_chain(chains[c][done:], blk, fr)
                else:
                    for e in chains[c][done:]:
                        e.transform_audio(blk, fr)
                how = rng.random()
                if plain:
                    w = ob[c].render(fr)
                    for e_ in ofx[c]:
                        e_.process(w)
                    if how < 0.85:
                        want[at:at + fr] = (0.0 if first else want[at:at + fr]) + w.sum(axis=2).T
                if how < 0.55 and not plain:
                    gpu_ctx.mix_deferred(blk, fr, E._Slice(bus, at), accumulate=not first)
                    first = False
                elif how < 0.85:
                    gpu_ctx.mix([blk], fr, E._Slice(bus, at), accumulate=not first)
                    first = False
                if rng.random() < 0.15:
                    seen.append(blk.download(fr).copy())
                if rng.random() < 0.7:
                    blk.release()
            if first:
                gpu_ctx.mix([], fr, E._Slice(bus, at))
            r = rng.random()
            if r < 0.08:
                seen.append(bus.download().copy())
            elif r < 0.14:
                v = float(rng.random())
                chains[0][2].control_set_param_by_index(T.CTL_FX_ATTENUATION, v)
                if plain:
                    newp = (T.FxParams * n)(*cur[0][2])
                    for i in range(n):
                        newp[i].attenuation = v
                    cur[0][2] = newp
                    ofx[0][2].set_params(newp)
            elif r < 0.16:
                which = int(rng.integers(2))
                for e in chains[which]:
                    e.reset()
                if plain:
                    ofx[which] = [oracle.Fx(k, cur[which][i]) for i, k in enumerate(kinds[which])]
            at += fr
        seen.append(bus.download().copy())
        if plain:
            seen.append(want)
        for c in range(2):
            for e in chains[c]:
                e.destroy()
            for blk in rots[c]:
                blk.destroy()
            synths[c].destroy()
        bus.destroy()
        gpu_ctx.fx_allpass_stream = False
        return seen

    sounding, seeds = 0, drawn_seeds(16)
    for seed in seeds:   # (a campaign of 1,000 seeds ran clean at the end of round 5)
        a, b = play(seed, False), play(seed, True)
        assert len(a) == len(b), seed
        sounding += int(np.abs(a[-1]).max() > 1e-3)   # (a reset late in a walk can leave the chains' delay lines silent to its end: seed 131)
        for k, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (seed, k)
        *_, ref, want = play(seed, False, plain=True)  # ... and the plain walk's bus (the forms' sums differ in order: 2e-6 of its scale)
        assert np.abs(a[-1].astype(np.float64) - ref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max())) * np.sqrt(192) / 8, seed
        # ... which follows the ORACLE's — the same voices through the same chains, parameter changes and resets, in f64: bus / voices RMS
        # <= 1e-5 of the larger of 1 and the bus's own peak (the chains have gain)
        err = np.sqrt(np.mean(((ref.astype(np.float64) - want) / 192) ** 2))
        assert err <= 1e-5 * max(1.0, float(np.abs(want).max()) / 192), (seed, err)
    assert sounding >= 0.9 * len(seeds)
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_allpass_stream_waits_that_time_out_lose_nothing(gpu_ctx):
    """The all-pass stream held up on purpose (a 0.4 s spin on it) and a 50 ms deadline: the host's waits for a block's release and
    for its render come back as errors; a caller that carries on — the streams still order themselves — gets the undisturbed bus."""
    from groove_amd import entities as E, lib, patches as P

    def run(disturb):
        assert not gpu_ctx.fx_allpass_stream
        gpu_ctx.fx_allpass_stream = True
        n = 256
        synth = E.WelshSynth(gpu_ctx, P.welsh_voices(n))
        fxp = (T.FxParams * n)(*[T.fx_params(cutoff_hz=800.0 + 7 * (i % 60), delay_seconds=0.04, attenuation=0.9, reverb_seconds=0.7) for i in range(n)])
        fx = [E.Effect(gpu_ctx, T.FX_DELAY, fxp), E.Effect(gpu_ctx, T.FX_REVERB, fxp)]
        rot = [gpu_ctx.block(n, 256) for _ in range(5)]
        blocks = 16
        bus = gpu_ctx.bus(blocks * 256)
        synth.handle_midi_events(P.note_on_all(n))
        errors = 0
        old = gpu_ctx.sync_timeout_ms
        try:
            for b in range(blocks):
                blk = rot[b % 5]
                if disturb and b == 7:
                    gpu_ctx.debug_spin(7, 400)          # side stream 7: the all-pass stream (the second bank stream)
                    gpu_ctx.sync_timeout_ms = 50
                for wait in (blk.wait_released, None, blk.wait_ready):
                    if wait is None:
                        synth.generate_batch_values_async(blk, 256)
                        continue
                    try:
                        wait()
                    except lib.GrooveError as e:
                        assert "not complete after 50 ms" in str(e), str(e)
                        errors += 1
                gpu_ctx.transform_chain(fx, blk, 256)
                gpu_ctx.mix_deferred(blk, 256, E._Slice(bus, b * 256), accumulate=False)
                blk.release()
        finally:
            gpu_ctx.sync_timeout_ms = old
        out = bus.download().copy()
        for e in fx:
            e.destroy()
        for blk in rot:
            blk.destroy()
        synth.destroy(); bus.destroy()
        gpu_ctx.fx_allpass_stream = False
        return out, errors

    a, ea = run(False)
    b, eb = run(True)
    assert ea == 0 and eb >= 1, (ea, eb)
    assert np.abs(a).max() > 1e-3 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_random_walks_of_the_fused_call_forms_against_the_plain_walk(oracle):
    """Seeded random call sequences over a mixed project (Welsh, FM and sampler banks of a few hundred to a few thousand voices): per
    block and bank one of the fused forms — groove_bank_render_mix, _paced, _deferred, the three banks in one launch, or an asynchronous
    render into a block and a (deferred) mix of it — with random note events, ragged blocks, downloads and event records in between.
    Every form must leave what the plain walk (render_mix, bank after bank) leaves: the forms' sums differ in order, so the bar is
    2e-6 of the bus's scale, which a skipped block, a lost reduction or a bank one block early misses by five orders.  The plain walk of the
    smaller projects is itself held to the f64 oracle playing the same events (bus / voices RMS <= 1e-5)."""
    from groove_amd import entities as E

    def play(seed, plain):
        rng = np.random.default_rng(seed)
        ctx = E.Context(0)
        try:
            n_sel = int(rng.choice([600, 3000, 9000]))
            # the Welsh bank's kernel form, the same in both walks: time-parallel (the default at these sizes), the all-kinds serial
            # kernel, the role-split kernel, or the per-kind pipelined kernels (the ABI's tuning knobs)
            form = str(rng.choice(["tp", "tp", "any", "split", "per-kind"]))
            if form != "tp":
                ctx.time_parallel_max_voices = 0
                ctx.split_max_waves = 1 << 20 if form == "split" else 0
                if form == "per-kind":
                    ctx.pipeline_min_waves = 1
            banks = _mixed_banks(ctx, np.arange(n_sel, dtype=np.int64), oracle if (plain and n_sel <= 3000) else None)
            insts = [b_[0] for b_ in banks]
            obs = [b_[2] for b_ in banks] if (plain and n_sel <= 3000) else None   # the oracle beside the plain walk (the small projects: it is a scalar loop)
            want = None
            long_blocks = rng.random() < 0.3     # calls of up to 4,096 frames (the time-parallel forms end at 256: the banks change kernels)
            cap = 4096 if long_blocks else 256
            rot = {id(inst): [ctx.block(inst.n, cap) for _ in range(3)] for inst in insts}
            blocks = 10 if long_blocks else 18
            frames_of = [int(rng.choice([256, 257, 1000, 4096, 4095, 100, 2048, 1] if long_blocks else [256, 256, 256, 256, 100, 37, 1])) for _ in range(blocks)]
            bus = ctx.bus(sum(frames_of))
            at = 0
            for b, fr in enumerate(frames_of):
                for inst in insts:   # random note events: a few voices on or off (strictly increasing voices: one event per voice and block)
                    k = int(rng.integers(0, 6))
                    if b == 0 or k:
                        voices = np.sort(rng.choice(inst.n, size=min(inst.n, 40 if b == 0 else k), replace=False)).astype(np.uint32)
                        keys = rng.integers(36, 84, size=voices.size).astype(np.uint8)
                        keys += (keys % 12 == 9).astype(np.uint8)       # (no A: docs/DSP_SPEC.md section 2, exact ties against the oracle)
                        ev = T.note_events_np(voices, keys, bool(b == 0 or rng.random() < 0.7))
                        inst.handle_midi_events(ev)
                        if obs:
                            obs[insts.index(inst)].note_events(ev)
                r1 = rng.random()
                one_launch = (not plain) and r1 < 0.2
                modes = [int(rng.integers(0, 5)) for _ in insts]   # (drawn in both walks: the sequences stay aligned)
                extra = rng.random()
                if plain:
                    for i, inst in enumerate(insts):
                        inst.render_mix(bus, fr, accumulate=i > 0, at_frame=at)
                    if obs:
                        w = sum(o_.render_bus(fr) for o_ in obs)
                        want = w if want is None else np.concatenate([want, w], axis=0)
                elif one_launch:
                    ctx.render_mix_banks_deferred(insts, bus, fr, accumulate=False, at_frame=at)
                else:
                    for i, inst in enumerate(insts):
                        m = modes[i]
                        if m == 0:
                            inst.render_mix(bus, fr, accumulate=i > 0, at_frame=at)
                        elif m == 1:
                            inst.render_mix_paced(bus, fr, accumulate=i > 0, at_frame=at)
                        elif m == 2:
                            inst.render_mix_deferred(bus, fr, accumulate=i > 0, at_frame=at)
                        else:
                            blk = rot[id(inst)][b % 3]
                            inst.generate_batch_values_async(blk, fr)
                            if m == 3:
                                ctx.mix([blk], fr, E._Slice(bus, at), accumulate=i > 0)
                            else:
                                ctx.mix_deferred(blk, fr, E._Slice(bus, at), accumulate=i > 0)
                            blk.release()
                if not plain:
                    if extra < 0.1:
                        bus.download()
                    elif extra < 0.2:
                        ev = ctx.event()
                        ctx.record(ev)
                        ctx.L.groove_event_destroy(ctx.h, ev)
                at += fr
            out = bus.download().astype(np.float64)
            assert ctx.debug_info()["zero_segments"] == 0
            if obs:   # the plain walk against the f64 oracle playing the same events: bus / voices RMS <= 1e-5 (SURVEY section 8d's bar)
                err = float(np.sqrt(np.mean(((out - want) / n_sel) ** 2)))
                assert err <= 1e-5, (seed, err)
            return out, n_sel
        finally:
            ctx.close()

    seeds = drawn_seeds(10)   # (a campaign of 300 seeds ran clean at the end of round 5)
    for seed in seeds:
        (want, n_sel), (got, _) = play(seed, True), play(seed, False)
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(want).max()) > 0.05, seed
        err = float(np.abs(got - want).max())
        assert err <= 2e-6 * scale * max(1.0, np.sqrt(n_sel) / 8), (seed, err, scale, n_sel)


def test_two_contexts_interleaved_on_one_device():
    """Two library contexts on one GPU — each with its own streams, buffers and deferred state — walking two projects in lock step, call by
    call (a mixed project through the fused forms on one, a chained instrument through the all-pass stream on the other): each leaves the
    bus it leaves alone, bit for bit.  Nothing in the library is process-wide except the RCCL handle."""
    from groove_amd import entities as E, projects as PJ

    def walk(ctxs, which):
        projs, buses = {}, {}
        for name in which:
            ctx = ctxs[name]
            projs[name] = PJ.Project(ctx, "mixed-131072", np.arange(3000, dtype=np.int64)) if name == "mixed" else PJ.Project(ctx, "chain-4096", np.arange(512, dtype=np.int64))
            buses[name] = ctx.bus(72 * 256)
        for b in range(72):   # (config #3's chain is silent for its first 15,435 frames: its delay lines)
            for name in which:
                projs[name].step(buses[name], b * 256)
        out = {name: buses[name].download().copy() for name in which}
        for name in which:
            projs[name].destroy(); buses[name].destroy()
        return out

    ctxs = {"mixed": E.Context(0), "chain": E.Context(0)}
    try:
        together = walk(ctxs, ("mixed", "chain"))
        alone = {**walk(ctxs, ("mixed",)), **walk(ctxs, ("chain",))}
        for name in ("mixed", "chain"):
            assert np.abs(alone[name]).max() > 1e-3
            assert np.array_equal(together[name].view(np.uint32), alone[name].view(np.uint32)), name
            assert ctxs[name].debug_info()["zero_segments"] == 0
    finally:
        for c in ctxs.values():
            c.close()


def test_two_contexts_on_two_threads():
    """The ABI's threading contract (include/groove_hip.h: a ctx and its handles belong to one thread at a time; different contexts are
    independent): two host threads, a context each, walking the same mixed project at the same time (ctypes releases the GIL inside every
    call) — both buses equal the bus of a lone walk, bit for bit."""
    import threading
    from groove_amd import entities as E, projects as PJ

    def walk(out, key):
        ctx = E.Context(0)
        try:
            proj = PJ.Project(ctx, "mixed-131072", np.arange(6000, dtype=np.int64))
            bus = ctx.bus(40 * 256)
            for b in range(40):
                proj.step(bus, b * 256)
            out[key] = (bus.download().copy(), ctx.debug_info()["zero_segments"])
            proj.destroy(); bus.destroy()
        except Exception as e:   # noqa: BLE001
            out[key] = e
        finally:
            ctx.close()

    res = {}
    walk(res, "alone")
    threads = [threading.Thread(target=walk, args=(res, k)) for k in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for k in ("alone", "a", "b"):
        assert not isinstance(res.get(k), Exception) and res.get(k) is not None, (k, res.get(k))
        assert res[k][1] == 0
    assert np.abs(res["alone"][0]).max() > 1e-2
    for k in ("a", "b"):
        assert np.array_equal(res[k][0].view(np.uint32), res["alone"][0].view(np.uint32)), k
