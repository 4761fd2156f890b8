#!/usr/bin/env python3
"""Where the time-parallel Welsh kernel's time goes: s_memtime ticks per wavefront and phase (welsh_tp.h, TP_PROBE).
Needs the measurement build (make -C groove_amd BUILD=build_tpprobe OUT=libvar_tpprobe.so EXTRA=-DGROOVE_TP_PROBE) in
place of libgroove_hip.so (tools/tp_probe.sh swaps it in on the GPU box).  Experiment tool, not part of bench.py.
    python3 tools/tp_probe.py [sizes...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, lib as L, projects as PJ  # noqa: E402

PHASES = ["params+state+idle_at", "noise", "env_seek+phases", "pass1 scans", "pass2 feed-forward+push", "affine scan",
          "passB filter out", "biquad head", "tile+barrier", "rows/block stores", "state store"]
sizes = [int(x) for x in sys.argv[1:]] or [256, 4096, 16384]
ctx = E.Context(0)
lib = L.load()
read = lib.groove_debug_tp_probe_read
read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
ctx.time_parallel_max_voices = 1 << 30
for n in sizes:
    proj = PJ.Project(ctx, "welsh-1m", np.arange(n, dtype=np.int64))
    bus = ctx.bus(84 * 256)
    proj.reset()
    for b in range(4):
        proj.step(bus, b * 256)
    ctx.synchronize()
    assert read(out, 1) == 0
    t0 = time.perf_counter()
    for b in range(4, 84):
        proj.step(bus, b * 256)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 80 * 1e3
    assert read(out, 0) == 0
    v = np.array(list(out), dtype=np.float64)
    waves = v[15]
    print(f"{n} voices: {ms:.4f} ms per block (probe build); {waves / 80:.0f} wavefronts per block; whole kernel {v[12] / waves:.0f} realtime ticks, "
          f"{v[:11].sum() / waves:.0f} memtime ticks per wavefront")
    for i, name in enumerate(PHASES):
        print(f"   {name:26s} {v[i] / waves:9.1f} ticks  {100 * v[i] / v[:11].sum():5.1f} %")
    proj.destroy(); bus.destroy()
ctx.close()
