"""The 18-block note-event script tools/event_minimise.py distilled from seed 19 of the random event test (voice 3, patch 3): replayed with
parts of the patch switched off, to see which component carries the deviation from the oracle.   python3 tools/retrigger_probe.py"""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

ctx = E.Context(0)
on, off = (lambda k: [(0, k, True)]), (lambda k: [(0, k, False)])
SCRIPT = [(100, on(70)), (256, []), (37, []), (100, []), (37, off(40)), (256, []), (256, on(91)), (256, []), (100, []), (37, []), (256, off(73)), (37, on(82)),
          (256, on(79)), (256, []), (256, []), (256, on(37)), (100, on(91)), (37, on(60)), (256, [])]


def dev(p, script=SCRIPT):
    params = (T.WelshParams * 8)(*[p] * 8)
    ob = O.Bank.welsh(params); s = E.WelshSynth(ctx, params); blk = ctx.block(8, 256)
    out = []
    for fr, evs in script:
        if evs:
            ob.note_events(T.note_events(evs)); s.handle_midi_events(T.note_events(evs))
        w = ob.render(fr); s.generate_batch_values(blk, fr); g = blk.download(fr).astype(np.float64)
        out.append(float(np.abs(g - w)[:, :, 0].max()))
    s.destroy(); blk.destroy()
    return out


def variant(**kw):
    p = P.welsh_patch(3)
    for k, v in kw.items():
        obj, _, field = k.rpartition("__")
        setattr(getattr(p, obj) if obj else p, field, v)
    return p


p0 = P.welsh_patch(3)
print("patch 3: amp env", p0.amp_envelope.attack, p0.amp_envelope.decay, p0.amp_envelope.sustain, p0.amp_envelope.release, "| filter env", p0.filter_envelope.attack, p0.filter_envelope.decay,
      p0.filter_envelope.sustain, p0.filter_envelope.release, "| lfo", p0.lfo_waveform, p0.lfo_routing, p0.lfo_frequency, p0.lfo_depth, "| cutoff", p0.filter_cutoff_hz, p0.filter_cutoff_start, p0.filter_cutoff_end,
      "| osc", p0.oscillator_1.waveform, p0.oscillator_2.waveform, p0.oscillator_2.tune, p0.oscillator_2_sync)
for name, p in (("as is", variant()), ("no sweep (cutoff_end 0)", variant(filter_cutoff_end=0.0)), ("no lfo", variant(lfo_routing=T.LFO_NONE)),
                ("filter env instant attack", variant(filter_envelope__attack=0.0)), ("amp env instant attack", variant(amp_envelope__attack=0.0)),
                ("filter env = amp env times", variant(filter_envelope__attack=p0.amp_envelope.attack, filter_envelope__decay=p0.amp_envelope.decay)),
                ("sine oscillators", variant(oscillator_1__waveform=T.WAVE_SINE, oscillator_2__waveform=T.WAVE_SINE)),
                ("filter attack 0.0601 s", variant(filter_envelope__attack=0.0601)), ("filter attack 0.0599 s", variant(filter_envelope__attack=0.0599)),
                ("filter attack 0.06002 s", variant(filter_envelope__attack=0.06002)), ("filter decay 0.31 s", variant(filter_envelope__decay=0.31))):
    d = dev(p)
    print(f"{name:28s} last four blocks {[f'{x:.1e}' for x in d[-4:]]}  max before {max(d[:-4]):.1e}")
