#!/usr/bin/env python3
"""Is a bench workload bound by host submission?  Times the K-step submission loop alone (no
synchronisation inside) and the loop plus the final synchronise (experiment tool).
    python3 tools/host_bound.py --workload chain-4096 [--steps 400]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
from groove_amd import projects as PJ  # noqa: E402
from groove_amd import entities as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="chain-4096")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--no-render-ahead", action="store_true")
    a = ap.parse_args()
    ctx = E.Context(0)
    V = PJ.WORKLOADS[a.workload]["voices"]
    proj = PJ.Project(ctx, a.workload, np.arange(V, dtype=np.int64), True, render_ahead=not a.no_render_ahead)
    bus = ctx.bus((a.steps + 8) * PJ.FRAMES)
    for s in range(8):
        proj.step(bus, s * PJ.FRAMES)
    ctx.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        proj.step(bus, (8 + s) * PJ.FRAMES)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"{a.workload}: submission {1e3 * (t1 - t0) / a.steps:.4f} ms/step, with final sync {1e3 * (t2 - t0) / a.steps:.4f} ms/step "
          f"(GPU still busy for {1e3 * (t2 - t1):.2f} ms after the last submission)")
    ctx.close()


if __name__ == "__main__":
    main()
