#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference holds NO golden vector for any oscillator / envelope / filter / effect output
(SURVEY.md §4), and its own source for them is absent, so these fixtures pin the oracle
(regression) and give the GPU tests a reference that does not need the oracle at run time.
Each file: inputs are regenerated from groove_amd.patches (closed-form, no RNG); outputs are the
oracle's f64 results stored as float32."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from groove_amd import patches as P, abi_types as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

FRAMES, BLOCK = 1024, 256


def fx_input(n, frames):
    t = np.arange(frames)[None, :, None]
    f = (110.0 * 2.0 ** (np.arange(n) % 37 / 12.0))[None, None, :]
    x = 0.5 * np.sin(2 * np.pi * f * t / 44100.0 + np.arange(2)[:, None, None])
    x[:, frames // 2:, :] = 0.0
    return x.astype(np.float32)


def fx_cases(n):
    def arr(**kw):
        a = (T.FxParams * n)()
        for i in range(n):
            a[i] = T.fx_params(**{k: (v[i] if isinstance(v, list) else v) for k, v in kw.items()})
        return a
    return {
        "gain": (T.FX_GAIN, arr(ceiling=[0.1 * (i + 1) for i in range(n)])),
        "bitcrusher": (T.FX_BITCRUSHER, arr(bits=[1 + 2 * i for i in range(n)])),
        "lp12": (T.FX_BIQUAD_LP12, arr(cutoff_hz=[200.0 * (i + 1) for i in range(n)], q=0.707)),
        "lp24": (T.FX_BIQUAD_LP24, arr(cutoff_hz=[300.0 * (i + 1) for i in range(n)], passband_ripple=0.8)),
        "chorus": (T.FX_CHORUS, arr(voices=4, delay_seconds=0.005)),
        "delay": (T.FX_DELAY, arr(delay_seconds=0.003)),
        "reverb": (T.FX_REVERB, arr(attenuation=0.95, reverb_seconds=1.25)),
    }


def main():
    out = {}
    n = 8
    # Welsh: patches 0, 5, 9, 13, 18, 22, 24, 31 cover every routing / waveform family
    sel = [0, 5, 9, 13, 18, 22, 24, 31]
    params = (T.WelshParams * n)(*[P.welsh_patch(j) for j in sel])
    keys = [48, 55, 60, 62, 67, 72, 76, 81]
    on = T.note_events([(v, keys[v], True) for v in range(n)])
    off = T.note_events([(v, keys[v], False) for v in range(n)])
    b = O.Bank.welsh(params)
    chunks = []
    for blk in range(FRAMES // BLOCK):
        if blk == 0:
            b.note_events(on)
        if blk == 2:
            b.note_events(off)
        chunks.append(b.render(BLOCK))
    out["welsh"] = np.concatenate(chunks, axis=1).astype(np.float32)
    fm = (T.FmParams * n)(*[P.fm_patch(j) for j in range(n)])
    b = O.Bank.fm(fm)
    chunks = []
    for blk in range(FRAMES // BLOCK):
        if blk == 0:
            b.note_events(on)
        if blk == 2:
            b.note_events(off)
        chunks.append(b.render(BLOCK))
    out["fm"] = np.concatenate(chunks, axis=1).astype(np.float32)
    x = fx_input(n, FRAMES)
    for name, (kind, p) in fx_cases(n).items():
        fx = O.Fx(kind, p)
        ys = [fx.process(np.ascontiguousarray(x[:, i:i + BLOCK, :]).astype(np.float64)) for i in range(0, FRAMES, BLOCK)]
        out["fx_" + name] = np.concatenate(ys, axis=1).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_r01.npz"), **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "golden_r01.npz")), "bytes")


if __name__ == "__main__":
    main()
