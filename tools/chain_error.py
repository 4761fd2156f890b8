#!/usr/bin/env python3
"""Where does the per-lane error of config #3's chain come from?  Per stage: signal RMS and error RMS vs the oracle
(64 sampled lanes of the 4,096-lane project, 60 blocks).  Experiment tool."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, abi_types as T, projects as PJ
from oracle import oracle as O

ctx = E.Context(0)
V, blocks, FR = 4096, 60, 256
spec = PJ.plan("chain-4096", np.arange(V))[0]
lanes = np.arange(64) * 64 + 17
synth = E.WelshSynth(ctx, spec["params"])
fx = [E.Effect(ctx, k, p) for k, p in spec["fx"]]
block = ctx.block(V, FR)
ob = O.Bank.welsh((T.WelshParams * 64)(*[spec["params"][int(i)] for i in lanes]))
ofx = [O.Fx(k, (T.FxParams * 64)(*[p[int(i)] for i in lanes])) for k, p in spec["fx"]]
keys = np.array([e.key for e in spec["events"][0]], dtype=np.uint8)
synth.handle_midi_events(spec["events"][0])
ob.note_events(T.note_events_np(np.arange(64, dtype=np.uint32), keys[lanes], True))
names = ["welsh", "lp12", "chorus", "delay", "reverb"]
stat = {n: [0.0, 0.0, 0.0] for n in names}
for b in range(blocks):
    synth.generate_batch_values(block, FR)
    want = ob.render(FR)
    stages = [(block.download(FR)[:, :, lanes].astype(np.float64), want.copy())]
    for e, oe in zip(fx, ofx):
        e.transform_audio(block, FR)
        oe.process(want)
        stages.append((block.download(FR)[:, :, lanes].astype(np.float64), want.copy()))
    for n, (g, w) in zip(names, stages):
        err = np.sqrt(np.mean((g - w) ** 2, axis=(0, 1)))
        sig = np.sqrt(np.mean(w ** 2, axis=(0, 1)))
        i = int(np.argmax(err))
        if err[i] > stat[n][0]:
            stat[n] = [float(err[i]), float(sig[i]), float(np.abs(w[:, :, i]).max()), b, int(lanes[i])]
for n in names:
    print(n, "worst lane err rms %.3e  (that lane: signal rms %.3e, peak %.3e, block %d, lane %d)" % tuple(stat[n]))
ctx.close()
