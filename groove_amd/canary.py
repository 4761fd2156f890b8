"""First-process canary for a fresh GPU box.

Observed on this pool (round 2, three times in ~25 jobs, each time in the FIRST GPU process a freshly leased MI355X box
ran, never in a later process on the same box, never in 30 back-to-back processes on a warm box): the million-voice
path — five streams at three priorities, one render kernel per Welsh base kind side by side — crawls.  The rocprofv3
trace of one such run (profiles/ has none of it: it never finished) shows ONE of the four render kernels, always the
one on the fourth normal-priority queue, starting tens of seconds after its siblings or running for 10-50 s instead of
0.36 ms, block after block, with the rest of the chip idle; every kernel of every other queue is normal.  A 176-block
run then takes 40 minutes.  Nothing in the library's own ordering explains it (no cross-queue wait is outstanding
while that kernel runs), and it cannot be provoked on demand.

What the measurement and test entry points do about it: before they touch the GPU themselves they run THIS module in a
child process — the same path for a few blocks — with a timeout.  If the child is the unlucky first process it is
killed when the timeout expires (its exact PID), and the parent, a later process on the box, runs normally.  It costs
about ten seconds; `GROOVE_NO_CANARY=1` skips it.
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
