import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or "/root/repo"
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O
import importlib.util
spec = importlib.util.spec_from_file_location("rpp", os.path.join(REPO, "tools", "random_patch_probe.py"))
src = open(os.path.join(REPO, "tools", "random_patch_probe.py")).read().replace("\nmain()\n", "\n")
ns = {}
exec(compile(src, "rpp", "exec"), ns)
seed, voice = int(sys.argv[1]), int(sys.argv[2])
n = 64
rng = np.random.default_rng(seed)
patches = [ns["random_patch"](rng) for _ in range(8)]
params = (T.WelshParams * n)(*[patches[(i // 8) % 8] for i in range(n)])
keys = rng.integers(30, 96, size=n).astype(np.uint8)
keys[keys % 12 == 9] += 1
lanes = np.arange(n, dtype=np.uint32)
ctx = E.Context(0)
ob = O.Bank.welsh(params); ob.note_events(T.note_events_np(lanes, keys, True))
s = E.WelshSynth(ctx, params); blk = ctx.block(n, 256); s.handle_midi_events(T.note_events_np(lanes, keys, True))
want, got = [], []
for b in range(40):
    if b == 24:
        ob.note_events(T.note_events_np(lanes, keys, False)); s.handle_midi_events(T.note_events_np(lanes, keys, False))
    want.append(ob.render(256)); s.generate_batch_values(blk, 256); got.append(blk.download(256))
want = np.concatenate(want, axis=1); got = np.concatenate(got, axis=1).astype(np.float64)
e = (got - want)[0, :, voice]
idx = np.argsort(-np.abs(e))[:12]
print("key", int(keys[voice]), "freq", 440 * 2 ** ((int(keys[voice]) - 69) / 12))
print("top errors (frame, err, got, want):", [(int(i), round(float(e[i]), 5), round(float(got[0, i, voice]), 5), round(float(want[0, i, voice]), 5)) for i in sorted(idx)])
print("count |e|>1e-4:", int((np.abs(e) > 1e-4).sum()), "of", e.size, " rms", float(np.sqrt(np.mean(e ** 2))))
same_patch = [v for v in range(n) if (v // 8) % 8 == (voice // 8) % 8]
print("same patch voices rms:", [(v, int(keys[v]), f"{float(np.sqrt(np.mean((got - want)[:, :, v] ** 2))):.1e}") for v in same_patch])
