"""One FRESH process on the multi-stream million-voice path (tools/stress_fresh.sh): 1,000,000 Welsh voices, a few blocks
through the per-kind pipelined kernels, the bus downloaded.  Prints one line:
    fresh: <voices> voices x <blocks> blocks in <ms> ms  crc <CRC-32 of the bus>  zero_segments <n>
Every process renders the same project from the same state, so every line of a stress run must carry the SAME crc."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(voices=1_000_000, blocks=8, device=0):
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
    crc = zlib.crc32(bus.download().tobytes())
    zeros = ctx.debug_info()["zero_segments"] + ctx.debug_info()["fast_table_misses"]   # (both counted assertions of csrc/diag.h)
    proj.destroy(); bus.destroy(); ctx.close()
    print(f"fresh: {voices} voices x {blocks} blocks in {dt * 1e3:.1f} ms  crc {crc:08x}  zero_segments {zeros}", flush=True)
    return 0 if zeros == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
