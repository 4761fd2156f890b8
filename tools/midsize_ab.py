"""Mid-size Welsh banks (the shards 2 - 8 GPUs see), the driver's window (blocks 5-24 from a reset state, 7 regions, median),
for one or several builds of the library inside ONE gpurun job:

    python3 tools/midsize_ab.py [--sizes 65536,125000,250000,500000] [--forms default,serial] [lib.so ...]

Every (library, size, form) is timed in its own process (the library is chosen at import: GROOVE_LIB_PATH).  `serial` forces the
all-kinds / per-kind serial kernels (no role split, no time-parallel form) through the ABI's tuning knobs.
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(voices, form, steps, warmup, repeats):
    sys.path.insert(0, REPO)
    import numpy as np
    import bench as B
    from groove_amd import entities as E, projects as PJ
    ctx = E.Context(0)
    ctx.sync_timeout_ms = 30000
    if form == "serial":
        ctx.split_max_waves = 0
        ctx.time_parallel_max_voices = 0
    proj = PJ.Project(ctx, "welsh-1m", np.arange(voices, dtype=np.int64))
    bus = ctx.bus((steps + warmup) * PJ.FRAMES)
    walls, kerns, _ = B.time_project(ctx, proj, bus, steps, warmup, repeats, True)
    ms = sorted(w / steps * 1e3 for w in walls)
    print(json.dumps({"voices": voices, "form": form, "ms_per_step": ms[len(ms) // 2], "min": ms[0], "all": [round(m, 4) for m in ms],
                      "kernel_form": [k[:60] for k in proj.kernel_forms()] if hasattr(proj, "kernel_forms") else None,
                      "zero_segments": ctx.debug_info()["zero_segments"]}))
    proj.destroy(); bus.destroy(); ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--sizes", default="65536,125000,250000,500000")
    ap.add_argument("--forms", default="default")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", nargs=2)
    args = ap.parse_args()
    if args.child:
        return child(int(args.child[0]), args.child[1], args.steps, args.warmup, args.repeats)
    libs = args.libs or [os.path.join(REPO, "groove_amd", "libgroove_hip.so")]
    for rnd in range(args.rounds):
        for size in (int(s) for s in args.sizes.split(",")):
            for form in args.forms.split(","):
                for lib in libs:
                    env = dict(os.environ, GROOVE_LIB_PATH=os.path.abspath(lib))
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(size), form, "--steps", str(args.steps),
                                        "--warmup", str(args.warmup), "--repeats", str(args.repeats)], env=env, capture_output=True, text=True, timeout=300)
                    line = (r.stdout.strip().splitlines() or ["{}"])[-1]
                    try:
                        d = json.loads(line)
                        print(f"{size:8d} {form:8s} {os.path.basename(lib):28s} {d['ms_per_step']:.4f} (min {d['min']:.4f})  zero {d['zero_segments']}", flush=True)
                    except Exception:
                        print(f"{size:8d} {form:8s} {os.path.basename(lib):28s} FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    sys.exit(main())
