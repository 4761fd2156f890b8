#!/usr/bin/env python3
"""Where the role-split Welsh kernel's step goes: cycles every role spends on its own work and at the workgroup barrier.
Needs the measurement build (make -C groove_amd BUILD=build_probe OUT=libvar_probe.so EXTRA=-DGROOVE_SPLIT_PROBE) in place
of libgroove_hip.so (tools/split_probe.sh does the swap on the GPU box).  Experiment tool, not part of bench.py.
    python3 tools/split_probe.py [--voices 65536] [--blocks 24] [--patches all|plain|f64|<ids>]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from groove_amd import abi_types as T  # noqa: E402
from groove_amd import entities as E  # noqa: E402
from groove_amd import lib as L  # noqa: E402
from groove_amd import patches as P  # noqa: E402
from patch_cost import bank_of  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=65536)
    ap.add_argument("--blocks", type=int, default=24)
    ap.add_argument("--patches", default="all")
    ap.add_argument("--no-probe", action="store_true", help="the product build: time only")
    a = ap.parse_args()
    routing = [P.welsh_patch(j).lfo_routing for j in range(P.N_PATCHES)]
    sets = {"all": list(range(P.N_PATCHES)),
            "plain": [j for j in range(P.N_PATCHES) if routing[j] not in (T.LFO_PITCH, T.LFO_PULSE_WIDTH)],
            "f64": [j for j in range(P.N_PATCHES) if routing[j] in (T.LFO_PITCH, T.LFO_PULSE_WIDTH)]}
    ids = sets.get(a.patches) or [int(x) for x in a.patches.split(",")]
    ctx = E.Context(0)
    lib = L.load()
    synth, on, off = bank_of(ctx, ids, a.voices)
    print("kernel form:", synth.kernel_form(256, True))
    form = synth.kernel_form(256, True)
    roles = 4 if "four wavefronts" in form else 3 if "three wavefronts" in form else 2
    if a.no_probe:
        read = lambda out, reset: 0  # noqa: E731
    else:
        read = getattr(lib, f"groove_debug_split_probe_read{roles}")
        read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    out = (C.c_ulonglong * 12)()
    bus = ctx.bus(256)
    e0, e1 = ctx.event(), ctx.event()
    synth.handle_midi_events(on)
    for b in range(4):
        synth.render_mix(bus, 256)
    ctx.synchronize()
    assert read(out, 1) == 0
    ctx.record(e0)
    for b in range(a.blocks):
        synth.render_mix(bus, 256)
    ctx.record(e1)
    ctx.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / a.blocks
    assert read(out, 0) == 0
    v = np.array(list(out), dtype=np.float64).reshape(4, 3)
    p0 = P.welsh_patch(ids[0])
    desc = f"o1={p0.oscillator_1.waveform} o2={p0.oscillator_2.waveform} lfo={p0.lfo_waveform} route={p0.lfo_routing} env_end={p0.filter_cutoff_end:.1f}" if len(ids) == 1 else ""
    print(f"{a.voices} voices, patches {a.patches}: {ms:.4f} ms per block ({'product' if a.no_probe else 'probe'} build), {roles} roles  {desc}")
    names = ["ctl  ", "osc  ", "mid  ", "back "] if roles == 4 else ["front", "mid  ", "back ", ""]
    for r, name in enumerate(names):
        if v[r, 2] == 0:
            continue
        busy, wait = v[r, 0] / v[r, 2] / a.blocks, v[r, 1] / v[r, 2] / a.blocks
        print(f"  role {name}: {busy:9.0f} ticks busy, {wait:9.0f} at the barrier per wavefront-block  ({busy / (busy + wait):.2f} busy)   [{int(v[r, 2] / a.blocks)} wavefronts]")
    ctx.close()


if __name__ == "__main__":
    main()
