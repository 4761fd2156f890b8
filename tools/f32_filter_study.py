"""The arithmetic-policy experiment of VERDICT round 4, item 4 (no GPU: the emulation tier, tests/emul): the 24 dB sections of every
synthetic benchmark patch as an fp32 transposed-direct-form-II recurrence instead of the product's f64 one, over the whole
172-block timeline (44,032 frames: note-on, note-off at block 86, release), per-voice RMS error against the f64 oracle.

    python3 tools/f32_filter_study.py            -> a markdown table (docs/DSP_SPEC.md holds the committed copy)

form 1 = plain fp32 coefficients (10 operations a frame), form 2 = the small-quantity form (14).  The lowest cutoff a voice's filter
can reach (static: the patch's cutoff; retuned: the sweep's lower end, clamped to 1 Hz) is known when the bank is uploaded.
"""
import math
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import patches as P, abi_types as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.emul import emul as E  # noqa: E402


def render(form, keys_shift=0):
    n = 32
    params = P.welsh_voices(n)
    E.lib().emul_set_f32_filter(form)
    bo, be = O.Bank.welsh(params), E.Bank.welsh(params)
    on, off = P.note_on_all(n), P.note_off_all(n)
    o, e = [], []
    for b in range(172):
        if b == 0:
            bo.note_events(on); be.note_events(on)
        if b == 86:
            bo.note_events(off); be.note_events(off)
        o.append(bo.render(256)); e.append(be.render(256))
    E.lib().emul_set_f32_filter(0)
    o, e = np.concatenate(o, axis=1), np.concatenate(e, axis=1).astype(np.float64)
    err = e - o
    return np.sqrt(np.mean(err ** 2, axis=(0, 1))), np.abs(err).max(axis=(0, 1)), np.sqrt(np.mean(o ** 2, axis=(0, 1)))


def cutoff_range(p):
    """The cutoffs (Hz) the voice's 24 dB filter takes (csrc/dsp_core.h welsh_frame_front, csrc/derive.h welsh_filter_f32_error)."""
    fc = lambda pct: 25.0 * 800.0 ** min(max(pct, 0.0), 1.0)  # noqa: E731
    if p.filter_cutoff_end != 0.0:
        a, b = fc(p.filter_cutoff_start), fc(p.filter_cutoff_start + (1.0 - p.filter_cutoff_start) * p.filter_cutoff_end)
    elif p.lfo_routing == T.LFO_FILTER_CUTOFF:
        a, b = fc(p.filter_cutoff_start * (1.0 - p.lfo_depth)), fc(p.filter_cutoff_start * (1.0 + p.lfo_depth))
    else:
        a = b = p.filter_cutoff_hz
    return min(a, b), min(max(a, b), 0.49 * 44100)


def render_kind():
    """The product's arithmetic for big banks: flagged patches in fp32 (tests/emul set_f32_kind), the rest in f64."""
    n = 32
    params = P.welsh_voices(n)
    bo, be = O.Bank.welsh(params), E.Bank.welsh(params)
    flagged = be.set_f32_kind(True)
    on, off = P.note_on_all(n), P.note_off_all(n)
    o, e = [], []
    for b in range(172):
        if b == 0:
            bo.note_events(on); be.note_events(on)
        if b == 86:
            bo.note_events(off); be.note_events(off)
        o.append(bo.render(256)); e.append(be.render(256))
    o, e = np.concatenate(o, axis=1), np.concatenate(e, axis=1).astype(np.float64)
    err = e - o
    return flagged, np.sqrt(np.mean(err ** 2, axis=(0, 1))), float(np.sqrt(np.mean((err.sum(axis=2) / n) ** 2)))


def main():
    import ctypes as C
    base, _, sig = render(0)
    f1, m1, _ = render(1)
    flagged, kind, bus = render_kind()
    rows = []
    for j in range(32):
        p = P.welsh_patch(j)
        lo, hi = cutoff_range(p)
        proxy = E.lib().emul_filter_f32_error(C.byref(p), 44100)
        rows.append((lo, j, hi, p.filter_passband_ripple, proxy, base[j], f1[j], m1[j], kind[j]))
    rows.sort()
    print("| patch | cutoff range (Hz) | ripple | host measurement (filter alone, fp32 vs f64) | fp32 filter allowed | voice RMS error, f64 filter (every other form) | voice RMS error, fp32 filter forced | its worst sample | voice RMS error, the per-kind kernels |")
    print("|---|---|---|---|---|---|---|---|---|")
    for lo, j, hi, rp, px, b, a1, w1, k in rows:
        rng = f"{lo:.0f}" if hi - lo < 0.5 else f"{lo:.0f} - {hi:.0f}"
        print(f"| {j} | {rng} | {rp:.2f} | {px:.1e} | {'yes' if px <= 2e-6 else 'no'} | {b:.1e} | {a1:.1e} | {w1:.1e} | {k:.1e} |")
    ok = [r for r in rows if r[4] <= 2e-6]
    print(f"\n{flagged} of 32 patches carry WF_FILTER_F32; their voices' worst RMS error {max(r[8] for r in ok):.1e} (f64 filter: {max(r[5] for r in ok):.1e}); "
          f"all 32 voices: {kind.max():.1e} (f64 filter everywhere: {base.max():.1e}); bus / 32 RMS {bus:.1e}.")
    print(f"fp32 filter forced on every patch: {sum(1 for r in rows if r[6] <= 5e-6)} of 32 stay <= 5e-6, {sum(1 for r in rows if r[6] <= 1e-5)} <= 1e-5, worst {f1.max():.1e} (patch {int(np.argmax(f1))}).")


if __name__ == "__main__":
    main()
