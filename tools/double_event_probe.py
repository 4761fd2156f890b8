import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or "/root/repo"
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O
ctx = E.Context(0)
for patch in (3, 0, 5):
  for name, second in (("single retrigger", [(0, 71, True)]), ("double note-on 60,71", [(0, 60, True), (0, 71, True)]), ("double 71,71", [(0, 71, True), (0, 71, True)]),
                       ("off then on", [(0, 60, False), (0, 71, True)]), ("on then off", [(0, 71, True), (0, 71, False)])):
    params = (T.WelshParams * 8)(*[P.welsh_patch(patch)] * 8)
    ob = O.Bank.welsh(params); s = E.WelshSynth(ctx, params); blk = ctx.block(8, 256)
    got, want = [], []
    for b in range(12):
        evs = [(0, 50, True)] if b == 0 else second if b == 6 else []
        if evs:
            ob.note_events(T.note_events(evs)); s.handle_midi_events(T.note_events(evs))
        want.append(ob.render(256)); s.generate_batch_values(blk, 256); got.append(blk.download(256))
    want = np.concatenate(want, axis=1); got = np.concatenate(got, axis=1).astype(np.float64)
    e = np.abs(got - want)[:, :, 0].max(axis=0)
    print(f"patch {patch} {name:22s} max err before {e[:6*256].max():.1e} after {e[6*256:].max():.1e}  signal after {np.abs(want[:, 6*256:, 0]).max():.2f}")
    s.destroy(); blk.destroy()
