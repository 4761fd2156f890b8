"""GPU tier: the fp32 filter kind (docs/DSP_SPEC.md section 11).  The FUSED per-kind kernels of big banks run the 24 dB filter of
patches the host has measured safe (WF_FILTER_F32) in fp32, through a second set of block bodies; every other kernel form keeps the
f64 recurrence.  Checked here on the 32 benchmark patches, one patch per workgroup (256 identical voices: the bank's bus / 256 is one
voice per patch), whole 172-block timeline with its note-off:
  * per-kind kernels (forced by the ABI's tuning knob) against the oracle, patch by patch: flagged patches <= 2e-6 RMS, all <= 1e-5;
  * against the all-kinds kernel (f64 filter for everybody): unflagged patches bit for bit, flagged patches not — they did take
    the other recurrence — and exactly the 19 patches the emulation tier's criterion names;
  * GROOVE_F32_FILTER=0 in a fresh process: the per-kind kernels' bus is the all-kinds kernel's, bit for bit, for all 32;
  * a voice can change kernel form between blocks (the fp32 state lives in the record's f64 fields): per-kind for 40 blocks, then
    time-parallel, stays within the bar."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from groove_amd import abi_types as T, patches as P

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER = 256
BLOCKS = 172


def _per_patch_buses(ctx, form, blocks=BLOCKS, switch_at=None):
    """One bank per patch (256 voices), fused render + mix; returns [32][frames][2] / 256."""
    from groove_amd import entities as E
    old = (ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves)
    out = []
    try:
        for j in range(P.N_PATCHES):
            ctx.time_parallel_max_voices, ctx.split_max_waves = 0, 0
            ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
            params = (T.WelshParams * PER)(*[P.welsh_patch(j)] * PER)
            s = E.WelshSynth(ctx, params)
            assert ("per base kind" in s.kernel_form(256, True)) == (form == "per-kind"), s.kernel_form(256, True)
            bus = ctx.bus(blocks * 256)
            key = np.full(PER, P.voice_keys(P.N_PATCHES)[j], dtype=np.uint8)      # the key the oracle's voice j plays
            lanes = np.arange(PER, dtype=np.uint32)
            s.handle_midi_events(T.note_events_np(lanes, key, True))
            for b in range(blocks):
                if b == 86:
                    s.handle_midi_events(T.note_events_np(lanes, key, False))
                if switch_at is not None and b == switch_at:
                    ctx.time_parallel_max_voices, ctx.pipeline_min_waves = old[0], old[2]
                s.render_mix(bus, 256, at_frame=b * 256)
            out.append(bus.download().astype(np.float64) / PER)
            s.destroy(); bus.destroy()
    finally:
        ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves = old
    return np.array(out)


def _oracle_voices(oracle, blocks=BLOCKS):
    params = P.welsh_voices(P.N_PATCHES)
    ob = oracle.Bank.welsh(params)
    ob.note_events(P.note_on_all(P.N_PATCHES))
    o = []
    for b in range(blocks):
        if b == 86:
            ob.note_events(P.note_off_all(P.N_PATCHES))
        o.append(ob.render(256))
    o = np.concatenate(o, axis=1)            # [2][frames][32]
    return np.transpose(o, (2, 1, 0))        # [32][frames][2]


def test_fp32_filter_kind_against_the_oracle_and_the_f64_form(gpu_ctx, oracle):
    from tests.emul import emul as EM
    flagged = np.array([EM.lib().emul_filter_f32_error(C.byref(P.welsh_patch(j)), 44100) <= 2e-6 for j in range(P.N_PATCHES)])
    assert flagged.sum() == 19
    want = _oracle_voices(oracle)
    kind = _per_patch_buses(gpu_ctx, "per-kind")
    f64 = _per_patch_buses(gpu_ctx, "all-kinds")
    rms_kind = np.sqrt(np.mean((kind - want) ** 2, axis=(1, 2)))
    rms_f64 = np.sqrt(np.mean((f64 - want) ** 2, axis=(1, 2)))
    assert np.sqrt(np.mean(want ** 2, axis=(1, 2))).min() > 1e-3
    assert rms_f64.max() <= 4e-6 and rms_kind.max() <= 1e-5, (rms_f64, rms_kind)
    assert rms_kind[flagged].max() <= 2e-6, rms_kind[flagged]
    same = np.array([np.array_equal(kind[j], f64[j]) for j in range(P.N_PATCHES)])
    assert np.array_equal(same, ~flagged), (same, flagged)       # exactly the flagged patches took the fp32 recurrence
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_a_voice_can_leave_the_fp32_form_between_blocks(gpu_ctx, oracle):
    want = _oracle_voices(oracle, 80)
    got = _per_patch_buses(gpu_ctx, "per-kind", blocks=80, switch_at=40)   # per-kind kernels, then the time-parallel form (f64)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(1, 2)))
    assert rms.max() <= 1e-5, rms


def test_switched_off_the_per_kind_kernels_give_the_f64_forms_bits():
    prog = (
        "import sys, zlib, numpy as np\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "sys.path.insert(0, %r)\n" % os.path.join(REPO, "tests") +
        "from groove_amd import entities as E\n"
        "import test_gpu_f32_filter as M\n"
        "ctx = E.Context(0)\n"
        "a = M._per_patch_buses(ctx, 'per-kind', blocks=24)\n"
        "b = M._per_patch_buses(ctx, 'all-kinds', blocks=24)\n"
        "print('SAME', int(sum(np.array_equal(a[j], b[j]) for j in range(32))))\n"
        "ctx.close()\n")
    for value, want in (("0", 32), ("1", 13)):   # 32 - 19 flagged patches
        r = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, GROOVE_F32_FILTER=value), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert f"SAME {want}" in r.stdout, (value, r.stdout[-300:])
