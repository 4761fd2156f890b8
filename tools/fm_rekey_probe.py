"""An FM voice re-triggered on ANOTHER key while it is still in its release (what a stolen voice of the host layer gets): GPU against the
oracle, both render forms, the 16 FM benchmark patches.   python3 tools/fm_rekey_probe.py"""
import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

ctx = E.Context(0)
n = 16
params = P.fm_voices(n)
lanes = np.arange(n, dtype=np.uint32)
old = ctx.time_parallel_max_voices
for form in ("time-parallel", "serial"):
    ctx.time_parallel_max_voices = old if form == "time-parallel" else 0
    for name, k1, k2, state in (("same key, from release", 60, 60, "release"), ("other key, from release", 41, 48, "release"), ("other key, while held", 41, 48, "held"),
                                ("other key, after the release", 41, 48, "idle")):
        s, ob = E.FmSynth(ctx, params), O.Bank.fm(params)
        blk = ctx.block(n, 256)
        worst = np.zeros(n)
        for b in range(140):
            evs = None
            if b == 0:
                evs = T.note_events_np(lanes, np.full(n, k1, dtype=np.uint8), True)
            elif b == 20 and state != "held":
                evs = T.note_events_np(lanes, np.full(n, k1, dtype=np.uint8), False)
            elif b == (30 if state != "idle" else 100):
                evs = T.note_events_np(lanes, np.full(n, k2, dtype=np.uint8), True)
            if evs is not None:
                s.handle_midi_events(evs); ob.note_events(evs)
            s.generate_batch_values(blk, 256)
            d = np.abs(blk.download(256).astype(np.float64) - ob.render(256)).max(axis=(0, 1))
            if b >= (30 if state != "idle" else 100):
                worst = np.maximum(worst, d)
        print(f"{form:14s} {name:30s} worst per patch after the re-trigger:", " ".join(f"{x:.0e}" for x in worst))
        s.destroy(); blk.destroy()
ctx.time_parallel_max_voices = old
