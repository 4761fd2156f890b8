import sys, time
sys.path.insert(0, '.')
import numpy as np
from groove_amd import entities as E, patches as P
ctx = E.Context(0)
for n in (125000, 32000, 1000000):
    params, vidx = P.welsh_voices_grouped(n)
    synth = E.WelshSynth(ctx, params)
    synth.handle_midi_events(P.grouped_note_events(vidx, True))
    bus = ctx.bus(256)
    for _ in range(8): synth.render_mix(bus, 256)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): synth.render_mix(bus, 256)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e3*(t1-t0)/200:.4f} ms/block (CPU), total {1e3*(t2-t0)/200:.4f} ms/block")
    synth.destroy(); bus.destroy()
