#!/usr/bin/env python3
"""Throughput of the fused Welsh path against the block length of one groove_bank_render_mix call
(the ABI takes up to 4096 frames): python3 tools/block_size.py [voices]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, patches as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = E.Context(0)
params, vidx = P.welsh_voices_grouped(n)
on = P.grouped_note_events(vidx, True)
for frames in (64, 256, 1024, 4096):
    synth = E.WelshSynth(ctx, params)
    synth.handle_midi_events(on)
    bus = ctx.bus(frames)
    total = 16384
    for _ in range(2):
        synth.render_mix(bus, frames)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(total // frames):
        synth.render_mix(bus, frames)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} voices, {frames:5d}-frame calls: {total / dt:10.0f} frames/s  ({dt / total * 256 * 1e3:.4f} ms per 256 frames)", flush=True)
    synth.destroy(); bus.destroy()
ctx.close()
