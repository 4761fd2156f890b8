"""Soak: every workload of bench.py rendered again and again from a reset state through the walks the bench uses (paced renders,
deferred reductions, the chain's paced walk), the bus of every repeat downloaded and its CRC-32 taken.  A workload's repeats render
the same project from the same state: ONE distinct CRC per workload is required, and the library's zero-segment counter must stay 0.

    python3 tools/soak.py [--seconds 60] [--blocks 12]
"""
import argparse
import collections
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0, help="per workload")
    ap.add_argument("--blocks", type=int, default=12)
    args = ap.parse_args()
    import numpy as np
    from groove_amd import entities as E, projects as PJ
    ctx = E.Context(0)
    ctx.sync_timeout_ms = 30000
    out = {}
    bad = False
    # (label, workload, voices, Project options): the five bench workloads, config #5's share of one of eight GPUs in ONE launch per
    # block (groove_banks_render_mix_deferred) and with its banks in turn (three row counts sharing the deferred row buffers)
    for label, workload, V, opts in (("welsh-1m", "welsh-1m", 0, {}), ("chain-4096", "chain-4096", 0, {}), ("mixed-131072", "mixed-131072", 0, {}),
                                     ("sampler-16384", "sampler-16384", 0, {}), ("welsh-256", "welsh-256", 0, {}),
                                     ("mixed-16384 one launch", "mixed-131072", 16384, {}), ("mixed-16384 banks in turn", "mixed-131072", 16384, {"one_launch": False})):
        V = V or PJ.WORKLOADS[workload]["voices"]
        proj = PJ.Project(ctx, workload, np.arange(V, dtype=np.int64), **opts)
        bus = ctx.bus(args.blocks * PJ.FRAMES)
        crcs = collections.Counter()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < args.seconds:
            proj.reset()
            bus.zero()
            for b in range(args.blocks):
                proj.step(bus, b * PJ.FRAMES)
            crcs[zlib.crc32(bus.download().tobytes())] += 1
            reps += 1
        proj.destroy(); bus.destroy()
        out[label] = {"repeats": reps, "distinct_bus_crcs": len(crcs), "crc": [f"{k:08x}" for k in crcs]}
        bad = bad or len(crcs) != 1
    ctx.synchronize()
    out["zero_segments"] = ctx.debug_info()["zero_segments"]
    ctx.close()
    print(json.dumps(out))
    return 1 if (bad or out["zero_segments"]) else 0


if __name__ == "__main__":
    sys.exit(main())
