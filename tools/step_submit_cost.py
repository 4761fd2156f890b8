#!/usr/bin/env python3
"""Host cost of submitting one project step against the time the GPU takes for it:
   tools/step_submit_cost.py [workload ...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, projects as PJ

ctx = E.Context(0)
for w in sys.argv[1:] or ["chain-4096", "sampler-16384", "welsh-256", "mixed-131072"]:
    V = PJ.WORKLOADS[w]["voices"]
    proj = PJ.Project(ctx, w, np.arange(V))
    K = 64
    bus = ctx.bus(K * PJ.FRAMES)
    for rep in range(3):
        proj.reset()
        for k in range(8): proj.step(bus, k * PJ.FRAMES)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(K): proj.step(bus, k * PJ.FRAMES)
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
    print(f"{w:16s} submit {1e3 * (t1 - t0) / K:.4f} ms/step (host)   total {1e3 * (t2 - t0) / K:.4f} ms/step", flush=True)
    proj.destroy(); bus.destroy()
ctx.close()
