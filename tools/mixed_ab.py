"""Config #5's per-GPU share (and other small mixed projects): ONE launch per block (groove_banks_render_mix_deferred,
csrc/welsh_tp.h tp_mixed_kernel) against the banks in turn, over the project's whole timeline in ONE gpurun job, every variant in
its own process.  (--vpw / --orders need a measurement build that reads GROOVE_MIXED_SAMPLER_VPW / GROOVE_MIXED_ORDER when the ctx
is created — round 5's experiment, results in profiles/r05_mixed_ab.log; the product build has the winners as constants.)

    python3 tools/mixed_ab.py [--voices 16384,4096,32768] [--vpw 2,4,8,16] [--rounds 2]
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(voices, one_launch, blocks, repeats):
    sys.path.insert(0, REPO)
    import zlib
    import numpy as np
    import bench as B
    from groove_amd import entities as E, projects as PJ
    ctx = E.Context(0)
    ctx.sync_timeout_ms = 30000
    proj = PJ.Project(ctx, "mixed-131072", np.arange(voices, dtype=np.int64), one_launch=one_launch)
    bus = ctx.bus(blocks * PJ.FRAMES)
    walls, kerns, _ = B.time_project(ctx, proj, bus, blocks, 0, repeats, True)
    crc = zlib.crc32(bus.download().tobytes())
    ms = sorted(w / blocks * 1e3 for w in walls)
    print(json.dumps({"voices": voices, "one_launch": bool(proj.one_launch), "ms_per_step": ms[len(ms) // 2], "min": ms[0], "all": [round(m, 4) for m in ms],
                      "kern_ms": sorted(kerns)[len(kerns) // 2], "bus_crc": crc}))
    proj.destroy(); bus.destroy(); ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", default="16384")
    ap.add_argument("--vpw", default="8")
    ap.add_argument("--orders", default="")
    ap.add_argument("--blocks", type=int, default=172)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", nargs=2)
    args = ap.parse_args()
    if args.child:
        return child(int(args.child[0]), args.child[1] == "1", args.blocks, args.repeats)
    for rnd in range(args.rounds):
        for v in (int(x) for x in args.voices.split(",")):
            variants = [("in turn", "0", {})] + [(f"one launch, sampler vpw {w}", "1", {"GROOVE_MIXED_SAMPLER_VPW": w}) for w in args.vpw.split(",")]
            variants += [(f"one launch, order {o}", "1", {"GROOVE_MIXED_ORDER": o}) for o in args.orders.split(",") if o]
            for label, one, extra in variants:
                env = dict(os.environ, **extra)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(v), one, "--blocks", str(args.blocks), "--repeats", str(args.repeats)],
                                   env=env, capture_output=True, text=True, timeout=300)
                try:
                    d = json.loads((r.stdout.strip().splitlines() or ["{}"])[-1])
                    print(f"{v:7d} {label:32s} {d['ms_per_step']:.4f} (min {d['min']:.4f}, events {d['kern_ms']:.4f})  crc {d['bus_crc']:08x}", flush=True)
                except Exception:
                    print(f"{v:7d} {label:32s} FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    sys.exit(main())
