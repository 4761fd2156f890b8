#!/usr/bin/env python3
"""Per-voice RMS error of the GPU path vs the f64 oracle over the config-#2 timeline (32 patches x 64
voices, grouped so the class-specialised uniform kernels run; materialised output).  Prints the
worst voices.  Test infrastructure (imports oracle/)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import entities as E, patches as P, abi_types as T
from oracle import oracle as O

ctx = E.Context(0)
n = 32 * 64
params, vidx = P.welsh_voices_grouped(n)
on, off = P.grouped_note_events(vidx, True), P.grouped_note_events(vidx, False)
synth = E.WelshSynth(ctx, params)
sample = np.arange(0, n, 64) + 3  # one voice per patch
ob = O.Bank.welsh((T.WelshParams * len(sample))(*[params[int(i)] for i in sample]))
keys = (36 + (7 * vidx[sample]) % 49).astype(np.uint8)
block = ctx.block(n, 256)
err2 = np.zeros(len(sample)); sig2 = np.zeros(len(sample)); frames = 0
for b in range(P.RENDER_BLOCKS):
    if b == 0:
        synth.handle_midi_events(on); ob.note_events(T.note_events_np(np.arange(len(sample), dtype=np.uint32), keys, True))
    if b == P.NOTE_OFF_FRAME // 256:
        synth.handle_midi_events(off); ob.note_events(T.note_events_np(np.arange(len(sample), dtype=np.uint32), keys, False))
    synth.generate_batch_values(block, 256)
    got = block.download(256)[:, :, sample].astype(np.float64)
    want = ob.render(256)
    err2 += ((got - want) ** 2).sum(axis=(0, 1)); sig2 += (want ** 2).sum(axis=(0, 1)); frames += 2 * 256
rms = np.sqrt(err2 / frames)
print("max per-voice RMS error %.3e (patch %d), median %.3e, signal RMS %.3f" % (rms.max(), rms.argmax(), np.median(rms), np.sqrt(sig2.mean() / frames)))
ctx.close()
