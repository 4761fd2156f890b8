"""One seed of tests/test_gpu_random_inputs.py::test_random_patches_at_other_sample_rates, taken apart: the worst voice's patch, its error per
block and per kernel form, its signal level.   python3 tools/random_sr_debug.py SEED"""
import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or "/root/repo"
sys.path.insert(0, REPO)
import ctypes
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O
O.build(ref=False)
seed = int(sys.argv[1])
n, blocks, off_at = 64, 30, 18
lanes = np.arange(n, dtype=np.uint32)
rng = np.random.default_rng(900 + seed)
sr = int(rng.choice([22050, 48000, 96000]))
patches = [P.random_welsh_patch(rng) for _ in range(8)]
params = (T.WelshParams * n)(*[patches[(i // 8) % 8] for i in range(n)])
keys = rng.integers(30, 96, size=n).astype(np.uint8)
keys[keys % 12 == 9] += 1


def play_oracle(rate):
    ob = O.Bank.welsh(params, sr=rate)
    ob.note_events(T.note_events_np(lanes, keys, True))
    want = []
    for b in range(blocks):
        if b == off_at:
            ob.note_events(T.note_events_np(lanes, keys, False))
        want.append(ob.render(256))
    return np.concatenate(want, axis=1)


def play_gpu(rate, form):
    ctx = E.Context(0)
    ctx.update_sample_rate(rate)
    if form != "tp":
        ctx.time_parallel_max_voices = 0
    ctx.split_max_waves = 0
    if form == "per-kind":
        ctx.pipeline_min_waves = 1
    s = E.WelshSynth(ctx, params); blk = ctx.block(n, 256)
    s.handle_midi_events(T.note_events_np(lanes, keys, True))
    got = []
    for b in range(blocks):
        if b == off_at:
            s.handle_midi_events(T.note_events_np(lanes, keys, False))
        s.generate_batch_values(blk, 256); got.append(blk.download(256))
    s.destroy(); blk.destroy(); ctx.close()
    return np.concatenate(got, axis=1).astype(np.float64)


want = play_oracle(sr)
print("seed", seed, "sr", sr)
for form in ("tp", "any", "per-kind"):
    got = play_gpu(sr, form)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    v = int(np.argmax(rms))
    sig = np.sqrt(np.mean(want[:, :, v] ** 2)); peak = np.abs(want[:, :, v]).max()
    print(form, "worst voice", v, "key", int(keys[v]), "rms err %.3e" % rms[v], "signal rms %.3f peak %.3f" % (sig, peak), "relative %.2e" % (rms[v] / sig))
    e = (got - want)[0, :, v].reshape(blocks, 256)
    print("  per-block rms err:", " ".join("%.1e" % x for x in np.sqrt(np.mean(e ** 2, axis=1))))
    print("  per-block signal peak:", " ".join("%.2f" % x for x in np.abs(want[0, :, v].reshape(blocks, 256)).max(axis=1)))
    print("  by voice:", " ".join("%.1e" % x for x in rms))
p = patches[(v // 8) % 8]
def dump(st, pre=""):
    for name, _ in st._fields_:
        x = getattr(st, name)
        if isinstance(x, ctypes.Structure): dump(x, pre + name + ".")
        elif hasattr(x, "__len__"): print("   ", pre + name, list(x))
        else: print("   ", pre + name, x)
dump(p)
# the same patch and keys at 44,100 Hz
want44 = play_oracle(44100); got44 = play_gpu(44100, "tp")
rms44 = np.sqrt(np.mean((got44 - want44) ** 2, axis=(0, 1)))
print("at 44100 by voice:", " ".join("%.1e" % x for x in rms44))
