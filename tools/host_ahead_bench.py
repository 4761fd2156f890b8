#!/usr/bin/env python3
"""Offline run of the C++ Orchestrator with and without render-ahead (experiment tool):
S sequenced Welsh synths, each through BiQuad -> Chorus -> Delay -> Reverb, 256-frame blocks.
    python3 tools/host_ahead_bench.py [--synths 8] [--voices 8] [--beats 16]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import abi_types as T  # noqa: E402
from groove_amd import patches as P  # noqa: E402
from groove_amd.host_binding import Orchestrator  # noqa: E402


def build(o, synths, voices, beats):
    seq = o.add_sequencer()
    for s in range(synths):
        w = o.add_welsh(P.welsh_patch(s % P.N_PATCHES), voices=voices)
        chain = [w, o.add_effect(T.FX_BIQUAD_LP12, T.fx_params(cutoff_hz=1000.0 + 100.0 * s, q=0.707)),
                 o.add_effect(T.FX_CHORUS, T.fx_params(voices=4, delay_seconds=0.25)),
                 o.add_effect(T.FX_DELAY, T.fx_params(delay_seconds=0.1)),
                 o.add_effect(T.FX_REVERB, T.fx_params(attenuation=0.95, reverb_seconds=1.25))]
        assert o.patch_chain_to_main_mixer(chain) == 0
        o.connect_midi_downstream(w, s % 16)
        for b in range(int(beats)):
            o.sequencer_insert(seq, s % 16, 48 + (5 * s + 3 * b) % 36, float(b), 0.75)
    o.sequencer_set_end(seq, beats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synths", type=int, default=8)
    ap.add_argument("--voices", type=int, default=8)
    ap.add_argument("--beats", type=float, default=16.0)
    a = ap.parse_args()
    for ahead in (False, True, False, True):
        o = Orchestrator(0, 44100, 128.0)
        o.set_render_ahead(ahead)
        build(o, a.synths, a.voices, a.beats)
        o.run(256)  # warm-up (allocations, first launches)
        t0 = time.perf_counter()
        out = o.run(256)
        dt = time.perf_counter() - t0
        print(f"render_ahead={ahead!s:5}  {len(out)} frames in {dt * 1e3:8.1f} ms  = {len(out) / dt:10.0f} frames/s  ({len(out) / dt / 44100:.2f}x RT)", flush=True)
        o.close()


if __name__ == "__main__":
    main()
