#!/usr/bin/env python3
"""Per-patch cost of the fused Welsh render: for each synthetic patch j (and for chosen subsets)
build a bank of N voices that all use that patch, play the config-#2 timeline (note-on block 0,
note-off block 86) and report ns per voice-block.  Experiment tool (not part of bench.py):
    python3 tools/patch_cost.py [--voices 262144] [--blocks 172]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import abi_types as T  # noqa: E402
from groove_amd import entities as E  # noqa: E402
from groove_amd import patches as P  # noqa: E402


def bank_of(ctx, patch_ids, n):
    """n voices, grouped: equal runs of each patch id."""
    table = (T.WelshParams * len(patch_ids))(*[P.welsh_patch(j) for j in patch_ids])
    size = C.sizeof(T.WelshParams)
    raw = np.frombuffer(bytes(bytearray(table)), dtype=np.uint8).reshape(len(patch_ids), size)
    sel = (np.arange(n) * len(patch_ids)) // n
    params = (T.WelshParams * n).from_buffer_copy(np.ascontiguousarray(raw[sel]).tobytes())
    keys = (36 + (7 * np.arange(n)) % 49).astype(np.uint8)
    on = T.note_events_np(np.arange(n, dtype=np.uint32), keys, True)
    off = T.note_events_np(np.arange(n, dtype=np.uint32), keys, False)
    return E.WelshSynth(ctx, params), on, off


def time_bank(ctx, synth, on, off, blocks):
    bus = ctx.bus(256)
    e0, e1 = ctx.event(), ctx.event()
    per_block = []
    for b in range(blocks):
        if b == 0:
            synth.handle_midi_events(on)
        elif b == P.NOTE_OFF_FRAME // 256:
            synth.handle_midi_events(off)
        ctx.record(e0)
        synth.render_mix(bus, 256)
        ctx.record(e1)
        ctx.synchronize()
        per_block.append(ctx.elapsed_ms(e0, e1))
    bus.destroy()
    return np.array(per_block)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=262144)
    ap.add_argument("--blocks", type=int, default=172)
    ap.add_argument("--subsets", default="", help="extra comma lists separated by ';' (e.g. '0,1;2,6')")
    ap.add_argument("--only", default="", help="comma list of single patches to time (default: all 32)")
    ap.add_argument("--serial", action="store_true", help="force the serial kernels (no role split, no time-parallel form)")
    a = ap.parse_args()
    ctx = E.Context(0)
    if a.serial:
        ctx.split_max_waves = 0
        ctx.time_parallel_max_voices = 0
    rows = []
    sets = [[j] for j in (map(int, a.only.split(",")) if a.only else range(P.N_PATCHES))]
    routing = [P.welsh_patch(j).lfo_routing for j in range(P.N_PATCHES)]
    plain = [j for j in range(P.N_PATCHES) if routing[j] not in (T.LFO_PITCH, T.LFO_PULSE_WIDTH)]
    f64 = [j for j in range(P.N_PATCHES) if routing[j] in (T.LFO_PITCH, T.LFO_PULSE_WIDTH)]
    sets += [plain, f64, list(range(P.N_PATCHES))]
    for s in a.subsets.split(";"):
        if s.strip():
            sets.append([int(x) for x in s.split(",")])
    for ids in sets:
        synth, on, off = bank_of(ctx, ids, a.voices)
        t = time_bank(ctx, synth, on, off, a.blocks)
        synth.destroy()
        p = P.welsh_patch(ids[0])
        label = ",".join(map(str, ids)) if len(ids) <= 4 else f"{len(ids)} patches"
        desc = ""
        if len(ids) == 1:
            desc = (f"o1={p.oscillator_1.waveform} o2={p.oscillator_2.waveform} lfo={p.lfo_waveform} route={p.lfo_routing} "
                    f"sync={p.oscillator_2_sync} env_end={p.filter_cutoff_end:.1f}")
        ns_vb = t.mean() * 1e6 / a.voices
        print(f"{label:>12}  mean {t.mean():7.4f} ms  first {t[1:5].mean():7.4f}  window {t[5:25].mean():7.4f}  sustain {t[60:80].mean():7.4f}  tail {t[-20:].mean():7.4f}"
              f"  {ns_vb:6.3f} ns/voice-block  {desc}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
