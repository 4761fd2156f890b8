"""Does the order in which a process first USES the library's side streams decide which hardware queue each lands on?  A fresh process per
order: the bank streams (6, 7, 8) and kind streams (0, 1, 2) are touched once in the given order, then config #3's paced walk is timed.
    for o in "6 7 8" "8 7 6" "7 6 8" "0 1 2 6 7 8" "8 7 6 2 1 0"; do python3 tools/stream_first_use_probe.py $o; done"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from groove_amd import entities as E, projects as PJ

order = [int(x) for x in sys.argv[1:]]
ctx = E.Context(0)
for k in order:
    ctx.L.groove_debug_spin(ctx.h, k, 0)
ctx.synchronize()
blocks = 172
proj = PJ.Project(ctx, "chain-4096", np.arange(4096, dtype=np.int64))
bus = ctx.bus(blocks * PJ.FRAMES)
res = []
for rep in range(4):
    proj.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for b in range(blocks):
        proj.step(bus, b * PJ.FRAMES)
    ctx.synchronize()
    res.append((time.perf_counter() - t0) / blocks * 1e3)
print("first use", order, " ms per block:", " ".join(f"{r:.4f}" for r in res), flush=True)
proj.destroy(); bus.destroy(); ctx.close()
