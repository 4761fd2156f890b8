#!/usr/bin/env python3
"""Per-block time of a Welsh bank by size, serial kernels against the time-parallel kernel (fused render+mix,
config-#2 voice rule and timeline blocks 4..83).  Experiment tool:  python3 tools/tp_bench.py [sizes...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, projects as PJ

sizes = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096, 8192, 16384, 32768, 65536]
ctx = E.Context(0)
default = ctx.time_parallel_max_voices
print(f"{'voices':>8} {'serial ms':>10} {'tp ms':>10}  ratio")
for n in sizes:
    res = []
    for form in (0, 1 << 30):
        ctx.time_parallel_max_voices = form
        proj = PJ.Project(ctx, "welsh-1m", np.arange(n, dtype=np.int64))
        bus = ctx.bus(84 * 256)
        best = 1e9
        for rep in range(3):
            proj.reset()
            for b in range(4):
                proj.step(bus, b * 256)
            ctx.synchronize()
            t0 = time.perf_counter()
            for b in range(4, 84):
                proj.step(bus, b * 256)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 80 * 1e3)
        res.append(best)
        proj.destroy(); bus.destroy()
    print(f"{n:8d} {res[0]:10.4f} {res[1]:10.4f}  {res[0] / res[1]:.2f}x")
ctx.time_parallel_max_voices = default
ctx.close()
