"""Is a workload's walk bound by the host?  Per step of the driver's window: wall time, and the time the calling thread spent
blocked inside the library's own waits (groove_debug_info: host_waits / host_waits_blocked / host_wait_ms).
    python3 tools/host_blocked.py [workload ...]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, projects as PJ  # noqa: E402

ctx = E.Context(0)
for w in sys.argv[1:] or ["mixed-131072", "chain-4096", "welsh-1m"]:
    V = PJ.WORKLOADS[w]["voices"]
    proj = PJ.Project(ctx, w, np.arange(V))
    K, W = 40, 5
    bus = ctx.bus((K + W) * PJ.FRAMES)
    for rep in range(4):
        proj.reset()
        for k in range(W):
            proj.step(bus, k * PJ.FRAMES)
        ctx.synchronize()
        d0 = ctx.debug_info()
        t0 = time.perf_counter()
        for k in range(K):
            proj.step(bus, (W + k) * PJ.FRAMES)
        t1 = time.perf_counter()
        d1 = ctx.debug_info()   # (waits for the ctx stream: counted below as the final wait)
        ctx.synchronize()
        t2 = time.perf_counter()
    blocked = (d1["host_wait_ms"] - d0["host_wait_ms"]) / K
    print(f"{w:14s} step {1e3 * (t2 - t0) / K:.4f} ms   submission loop {1e3 * (t1 - t0) / K:.4f} ms/step, of which blocked in waits {blocked:.4f} "
          f"({d1['host_waits'] - d0['host_waits']} waits, {d1['host_waits_blocked'] - d0['host_waits_blocked']} found the device busy) "
          f"-> host busy {1e3 * (t1 - t0) / K - blocked:.4f} ms/step", flush=True)
    proj.destroy(); bus.destroy()
ctx.close()
