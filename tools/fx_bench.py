#!/usr/bin/env python3
"""Time one effect kind on an n-lane block, serial form against the time-parallel form (experiment tool)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, abi_types as T

ctx = E.Context(0)
default = ctx.time_parallel_max_voices
for kind, name in ((T.FX_BIQUAD_LP12, "biquad"), (T.FX_BIQUAD_LP24, "lp24")):
    for n in (1, 64, 1024, 4096, 16384, 65536, 262144):
        params = (T.FxParams * n)(*[T.fx_params(cutoff_hz=500.0 + 10 * (i % 64)) for i in range(n)])
        blk = ctx.block(n, 256)
        blk.upload(np.random.default_rng(1).standard_normal((2, 256, n)).astype(np.float32) * 0.1)
        res = []
        for form in (0, default):
            ctx.time_parallel_max_voices = form
            fx = E.Effect(ctx, kind, params)
            for _ in range(20): fx.transform_audio(blk, 256)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): fx.transform_audio(blk, 256)
            ctx.synchronize()
            res.append((time.perf_counter() - t0) / 200 * 1e6)
            fx.destroy()
        print(f"{name:7s} n={n:6d}: serial {res[0]:7.1f} us   time-parallel {res[1]:7.1f} us")
        blk.destroy()
ctx.time_parallel_max_voices = default
ctx.close()
