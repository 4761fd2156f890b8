"""config #3's step against the side stream its bank renders on: k dummy banks are created first, so the chain's bank takes the
(k mod 3)-th bank stream, with the all-pass stream on and off.  python3 tools/ap_slot_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groove_amd import entities as E, patches as P, projects as PJ

ctx = E.Context(0)
blocks = 172
for k in range(7):
    dummies = [E.WelshSynth(ctx, P.welsh_voices(64)) for _ in range(k)]
    for ap in (True, False):
        proj = PJ.Project(ctx, "chain-4096", np.arange(4096, dtype=np.int64), allpass_stream=ap)
        bus = ctx.bus(blocks * PJ.FRAMES)
        best = 1e9
        for rep in range(3):
            proj.reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            for b in range(blocks):
                proj.step(bus, b * PJ.FRAMES)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / blocks * 1e3)
        print(f"dummy banks {k}  all-pass stream {int(ap)}: {best:.4f} ms per block", flush=True)
        proj.destroy(); bus.destroy()
    for d in dummies:
        d.destroy()
ctx.close()
