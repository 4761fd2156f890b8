"""The stall of DESIGN.md section 7, hunted inside ONE process.

Every iteration is what a timed region of bench.py starts with: reset the bank, note-on for every voice, a few blocks
of the million-voice project through the per-kind pipelined kernels — the window in which an instant-attack voice's
envelope record goes from (ATTACK, n 0, N 0) to a plateau, and in which a padding wave's shadow lanes could load it torn.
After every iteration the bus of the rendered blocks is downloaded and its CRC-32 recorded: all iterations of a run
render the same project from the same state, so there must be exactly ONE distinct CRC (a voice whose state record had
been corrupted would change it).  At the end the library's counters (groove_debug_info) are printed:

  product build                        zero_segments must be 0
  -DGROOVE_DIAG_SHADOW_IN_MIN build    shadow_zero_waves > 0 is the round-3 stall, counted instead of spinning, and
                                       `records` says who: active 0, count 0 (a padding wave), a plateau with N = 0

  GROOVE_LIB_PATH=groove_amd/libgroove_diag_shadow.so python3 tools/zero_segment_hunt.py --iters 2000
  python3 tools/zero_segment_hunt.py --iters 2000                      # the product build
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
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--voices", type=int, default=1_000_000)
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    import numpy as np
    from groove_amd import entities as E, projects as PJ, lib
    ctx = E.Context(0)
    ctx.sync_timeout_ms = 30000
    proj = PJ.Project(ctx, "welsh-1m", np.arange(args.voices, dtype=np.int64))
    form = proj.dominant.kernel_form(PJ.FRAMES, True)
    bus = ctx.bus(args.blocks * PJ.FRAMES)
    crcs = collections.Counter()
    t0 = time.perf_counter()
    for it in range(args.iters):
        proj.reset()
        for b in range(args.blocks):
            proj.step(bus, b * PJ.FRAMES)
        crcs[zlib.crc32(bus.download().tobytes())] += 1
    dt = time.perf_counter() - t0
    ctx.synchronize()
    info = ctx.debug_info()
    out = {"label": args.label, "library": os.path.basename(lib.LIB_PATH), "kernel_form": form, "voices": args.voices, "iterations": args.iters,
           "blocks_per_iteration": args.blocks, "seconds": round(dt, 1), "distinct_bus_crcs": len(crcs),
           "crc_counts": {f"{k:08x}": v for k, v in crcs.most_common(4)},
           "counters": {k: info[k] for k in ("zero_segments", "diag_build", "shadow_zero_lanes", "shadow_zero_waves", "records") if k in info}}
    print(json.dumps(out))
    proj.destroy(); bus.destroy(); ctx.close()
    return 0 if (len(crcs) == 1 and info["zero_segments"] == 0) else 1


if __name__ == "__main__":
    sys.exit(main())
