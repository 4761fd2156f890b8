"""Canary process for the GPU test session and smoke().

Observed on this pool (round 2, four times in ~60 processes that ran the million-voice path on recently leased MI355X
boxes; never in 30 back-to-back processes on a warm box): the path — five streams at three priorities, one render
kernel per Welsh base kind side by side — crawls for the life of the process.  The rocprofv3 trace of one such run
shows ONE of the four render kernels, always the one on the fourth normal-priority queue, starting tens of seconds
after its siblings or running for 10-50 s instead of 0.36 ms, block after block, with the rest of the chip idle and
every kernel of every other queue normal.  A 176-block run then takes 40 minutes.  The next process on the same box is
fine.  Three of the four were the first GPU process of their box, one the second.  Nothing in the library's own
ordering explains it (no cross-queue wait is outstanding while that kernel runs) and it cannot be provoked on demand.

`bench.py` protects its measurement with a watchdog (the measurement runs in a child process that is killed and
restarted if it does not finish in time).  A test session cannot restart itself, so it does the next best thing:
before it touches the GPU it runs THIS module — the same path for eight blocks — in a child process with a timeout,
which takes the most exposed position (a box's first GPU process) and is killed (its exact PID) if it crawls.
About ten seconds; `GROOVE_NO_CANARY=1` skips it.
"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(voices=1_000_000, blocks=8, device=0):
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import numpy as np
    from groove_amd import entities as E, projects as PJ
    ctx = E.Context(device)
    proj = PJ.Project(ctx, "welsh-1m", np.arange(voices, dtype=np.int64))
    bus = ctx.bus(blocks * PJ.FRAMES)
    t0 = time.perf_counter()
    for b in range(blocks):
        proj.step(bus, b * PJ.FRAMES)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    proj.destroy(); bus.destroy(); ctx.close()
    print(f"canary: {voices} voices x {blocks} blocks in {dt * 1e3:.1f} ms", flush=True)


def run(timeout_s=75.0, device=0):
    """Run the canary in a child process; returns 'ok', 'killed' (timed out: the child was the unlucky one), 'skipped'
    or 'failed' (any other non-zero exit: the real run will report the error itself)."""
    if os.environ.get("GROOVE_NO_CANARY") == "1":
        return "skipped"
    # never start a child from under a profiler: rocprofv3's preloaded tool would follow it (and a child launched from under
    # --pmc is the forbidden re-launch of DESIGN.md section 7)
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_")) for k in os.environ):
        return "skipped (profiler)"
    env = dict(os.environ, GROOVE_NO_CANARY="1", GROOVE_CANARY_DEVICE=str(device))
    p = subprocess.Popen([sys.executable, "-m", "groove_amd.canary"], cwd=REPO, env=env,
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        rc = p.wait(timeout=timeout_s)
        return "ok" if rc == 0 else "failed"
    except subprocess.TimeoutExpired:
        p.kill()          # this child only
        p.wait()
        time.sleep(2.0)   # let the driver tear its queues down
        return "killed"


if __name__ == "__main__":
    main(device=int(os.environ.get("GROOVE_CANARY_DEVICE", "0")))
