"""GPU tier: the time-parallel Welsh kernel (csrc/welsh_tp.h: one wavefront per voice, lanes = time) against the
serial kernels (same bank, groove_set_time_parallel_max_voices(0)) and the oracle — materialised and fused forms,
ragged block lengths, note-off, idle tails, re-trigger, every kind of patch the kernel treats specially (pitch LFO:
prefix sums; hard sync: max-scan; noise: serial pre-pass; resonance / cutoff LFO; envelope-retuned filter), and the
voice state after every block (bit-for-bit apart from the filter's four f64 values)."""
import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T

pytestmark = pytest.mark.gpu


def _patches():
    pats = [P.welsh_patch(j) for j in range(32)]
    for k, r in enumerate([T.LFO_PITCH_OSC2, T.LFO_PW_OSC1, T.LFO_PW_OSC2, T.LFO_RESONANCE, T.LFO_CUTOFF_AMP]):
        p = P.welsh_patch(4 + 5 * k)
        p.oscillator_1.waveform, p.oscillator_1.duty = T.WAVE_PULSE_WIDTH, 0.3
        p.lfo_routing, p.lfo_depth = r, 0.2
        if r == T.LFO_CUTOFF_AMP:
            p.filter_cutoff_end = 0.0
        pats.append(p)
    q = P.welsh_patch(1); q.oscillator_2_sync = 1; q.lfo_routing = T.LFO_PITCH; q.lfo_waveform = T.WAVE_TRIANGLE; pats.append(q)
    q = P.welsh_patch(5); q.lfo_waveform = T.WAVE_NOISE; q.lfo_routing = T.LFO_AMPLITUDE; pats.append(q)
    q = P.welsh_patch(2); q.oscillator_1.waveform = T.WAVE_NOISE; q.oscillator_2_sync = 1; pats.append(q)
    return pats


@pytest.mark.parametrize("vpw", [1, 2])
def test_time_parallel_matches_serial_kernels_and_oracle(gpu_ctx, oracle, vpw):
    """vpw = 2: the two-voices-per-wavefront form (welsh_tp_kernel<.., VPW = 2>: 32 lanes x 8 frames per voice) on a bank whose
    adjacent voices share a patch — forced for this small bank by groove_set_time_parallel_pair_min_voices(1); the bank's size
    is odd, so the last wavefront holds one voice."""
    from groove_amd import entities as E
    pats = _patches()
    if vpw == 1:
        n = len(pats) * 3        # three voices per patch, different keys; not a multiple of the 4 voices per workgroup
        params = (T.WelshParams * n)(*[pats[i % len(pats)] for i in range(n)])
    else:
        n = len(pats) * 2 + 1    # voices 2k and 2k + 1 on patch k; not a multiple of the 8 voices per workgroup
        params = (T.WelshParams * n)(*[pats[(i // 2) % len(pats)] for i in range(n)])
    old_pair = gpu_ctx.time_parallel_pair_min_voices
    gpu_ctx.time_parallel_pair_min_voices = 1 if vpw == 2 else 0
    keys = (38 + (7 * np.arange(n)) % 40).astype(np.uint8)
    keys[keys % 12 == 9] += 1    # no A notes: exact edge ties (DSP_SPEC §2)
    on = T.note_events_np(np.arange(n, dtype=np.uint32), keys, True)
    off = T.note_events_np(np.arange(n, dtype=np.uint32), keys, False)
    old = gpu_ctx.time_parallel_max_voices
    assert old >= n
    tp, ser = E.WelshSynth(gpu_ctx, params), E.WelshSynth(gpu_ctx, params)
    assert ("two voices per wavefront" in tp.kernel_form(256, False)) == (vpw == 2), tp.kernel_form(256, False)
    orc = oracle.Bank.welsh(params)
    block = gpu_ctx.block(n, 256)
    worst_ser = worst_orc = 0.0
    try:
        for blk in range(60):
            if blk in (0, 40):
                for b in (tp, ser): b.handle_midi_events(on)
                orc.note_events(on)
            if blk == 18:
                for b in (tp, ser): b.handle_midi_events(off)
                orc.note_events(off)
            frames = [256, 256, 100, 7, 1, 255][blk % 6]
            gpu_ctx.time_parallel_max_voices = old
            tp.generate_batch_values(block, frames)
            a = block.download(frames).astype(np.float64)
            st_tp = tp.download_state()
            gpu_ctx.time_parallel_max_voices = 0
            ser.generate_batch_values(block, frames)
            b_ = block.download(frames).astype(np.float64)
            st_ser = ser.download_state()
            c = orc.render(frames)
            scale = np.maximum(1.0, np.abs(c).max(axis=(0, 1)))
            worst_ser = max(worst_ser, float((np.abs(a - b_).max(axis=(0, 1)) / scale).max()))
            worst_orc = max(worst_orc, float((np.sqrt(np.mean((a - c) ** 2, axis=(0, 1))) / scale).max()))
            # state (WelshState words: 3 x OscState {u64 phase, x1, x2}, 2 x u64 increments, 2 x EnvState, 4 x f64 filter,
            # vflags, pad): noise generators, increments, envelopes, flags bit for bit; oscillator phases to 2^-40 turns
            # (the serial smooth-LFO kinds advance the pitch factor by a recurrence, this kernel evaluates it exactly:
            # increments differ in their last bits); the filter to f64 rounding
            exact = [2, 3, 6, 7, 10, 11] + list(range(12, 30)) + [38]
            assert np.array_equal(st_tp[exact], st_ser[exact]), blk
            for w in (0, 4, 8):
                pa = st_tp[w].astype(np.uint64) | (st_tp[w + 1].astype(np.uint64) << np.uint64(32))
                pb = st_ser[w].astype(np.uint64) | (st_ser[w + 1].astype(np.uint64) << np.uint64(32))
                d = (pa - pb).astype(np.int64)
                assert np.abs(d).max() <= 1 << 24, (blk, w)
            f_tp = st_tp[30:38].T.copy().view(np.float64); f_ser = st_ser[30:38].T.copy().view(np.float64)
            assert np.max(np.abs(f_tp - f_ser) / np.maximum(1e-3, np.abs(f_ser))) <= 1e-6, blk
    finally:
        gpu_ctx.time_parallel_max_voices = old
        gpu_ctx.time_parallel_pair_min_voices = old_pair
    assert worst_ser <= 2e-6, worst_ser
    assert worst_orc <= 1e-5, worst_orc
    tp.destroy(); ser.destroy(); block.destroy()


def test_time_parallel_fused_bus_config2(gpu_ctx, oracle):
    """Config #2 (256 voices, 172 blocks) fused render+mix through the time-parallel kernel against the oracle bus."""
    from groove_amd import projects as PJ
    from oracle.projects import OracleProject
    assert gpu_ctx.time_parallel_max_voices >= 256
    sel = np.arange(256)
    proj = PJ.Project(gpu_ctx, "welsh-256", sel)
    bus = gpu_ctx.bus(172 * 256)
    for b in range(172):
        proj.step(bus, b * 256)
    got = bus.download().astype(np.float64) / 256
    want = OracleProject("welsh-256", sel).render(172) / 256
    proj.destroy(); bus.destroy()
    assert np.sqrt(np.mean(want ** 2)) > 1e-3
    assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6


def test_two_voices_per_wavefront_fused_and_chained_forms(gpu_ctx, oracle):
    """Config #3's shape at a size the oracle renders in seconds: 256 Welsh voices laid out synth by synth (8 per patch, so
    adjacent voices share a patch), the BiQuad -> Chorus -> Delay -> Reverb chain with its IIR head fused into the render, the
    time-parallel kernel forced to two voices per wavefront.  Bus against the oracle; and the fused render + mix of a plain
    Welsh project in the same form against the one-voice form (the forms differ by f64 rounding of the filter scan only)."""
    from groove_amd import projects as PJ
    from oracle.projects import OracleProject
    old_pair = gpu_ctx.time_parallel_pair_min_voices
    try:
        sel = np.arange(256)
        blocks = 80  # (the chain's 11,025-frame delay line is wet only: the bus is silent for the first 43 blocks)
        buses = {}
        for pair_min in (0, 1):
            gpu_ctx.time_parallel_pair_min_voices = pair_min
            proj = PJ.Project(gpu_ctx, "chain-4096", sel)
            forms = [inst.kernel_form(256, False) for inst, _, _, _ in proj.banks]
            bus = gpu_ctx.bus(blocks * 256)
            for b in range(blocks):
                proj.step(bus, b * 256)
            buses[pair_min] = bus.download().astype(np.float64) / 256
            proj.destroy(); bus.destroy()
            if forms:
                assert any("two voices per wavefront" in f for f in forms) == (pair_min == 1), forms
        want = OracleProject("chain-4096", sel).render(blocks) / 256
        sig = np.sqrt(np.mean(want ** 2))
        assert sig > 1e-3
        for k, got in buses.items():
            assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-5 * max(1.0, sig / 0.1), k
        assert np.max(np.abs(buses[0] - buses[1])) <= 2e-6 * max(1.0, np.abs(want).max())
        fused = {}
        for pair_min in (0, 1):
            gpu_ctx.time_parallel_pair_min_voices = pair_min
            proj = PJ.Project(gpu_ctx, "welsh-1m", sel)
            bus = gpu_ctx.bus(100 * 256)
            for b in range(100):
                proj.step(bus, b * 256)
            fused[pair_min] = bus.download().astype(np.float64) / 256
            proj.destroy(); bus.destroy()
        want = OracleProject("welsh-1m", sel).render(100) / 256
        for k, got in fused.items():
            assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6, k
        assert np.max(np.abs(fused[0] - fused[1])) <= 1e-6
    finally:
        gpu_ctx.time_parallel_pair_min_voices = old_pair
