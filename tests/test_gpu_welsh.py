"""GPU parity, Welsh voices: HIP path through the
C ABI vs the f64 oracle on the same seeded inputs.

Tolerances (fp32 device output vs f64 oracle):
  per voice:  RMS error <= 1e-5 (signal RMS ~0.1-0.7), i.e. the north-star bar per voice;
  bus / V:    RMS error <= 1e-5 (north star), expected ~1e-7.
"""
import numpy as np
import pytest

from groove_amd import patches as P, types as T

pytestmark = pytest.mark.gpu

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
