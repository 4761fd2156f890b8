"""What a caller pays who wants every block's bus ON THE HOST before it submits the next block (the reference's own loop shape:
Orchestrator::tick fills a host buffer per call, orchestrator.rs:856-877): the driver's window with a synchronous 2 KiB download of
the block's bus after every step, beside the resident form bench.py reports.  Never part of `value`.
    python3 tools/pcie_inclusive.py [workload ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, projects as PJ  # noqa: E402

ctx = E.Context(0)
K, W = 20, 5
for w in sys.argv[1:] or ["welsh-1m", "mixed-131072", "chain-4096", "welsh-256"]:
    V = PJ.WORKLOADS[w]["voices"]
    proj = PJ.Project(ctx, w, np.arange(V))
    bus = ctx.bus((K + W) * PJ.FRAMES)
    host = np.empty((PJ.FRAMES, 2), dtype=np.float32)
    res = {}
    for mode in ("resident", "bus to the host every block"):
        ts = []
        for rep in range(5):
            proj.reset()
            for k in range(W):
                proj.step(bus, k * PJ.FRAMES)
            ctx.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                proj.step(bus, (W + k) * PJ.FRAMES)
                if mode != "resident":
                    E._lib.check(ctx.L.groove_download(ctx.h, bus.at((W + k) * PJ.FRAMES), host.ctypes.data_as(C.POINTER(C.c_float)), PJ.FRAMES * 2), ctx.h)
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    print(f"{w:14s} resident {res['resident']:.4f} ms per block; with the block's bus downloaded synchronously after every step {res['bus to the host every block']:.4f}", flush=True)
    proj.destroy(); bus.destroy()
ctx.close()
