"""tests/test_gpu_random_inputs.py::test_random_note_event_sequences_in_every_kernel_form by hand for one seed: per form the worst voice, where
its error sits and the events that voice received.   python3 tools/random_events_debug.py <seed>"""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

seed = int(sys.argv[1])
n, blocks = 96, 36
params = P.welsh_voices(n)
rng = np.random.default_rng(seed)
script, sizes = [], []
for b in range(blocks):
    evs = []
    for _ in range(int(rng.integers(0, 13)) if b else 0):
        voice = T.ALL_VOICES if rng.random() < 0.06 else int(rng.integers(n)) if rng.random() < 0.7 else int(rng.integers(4))
        key = int(rng.integers(30, 96))
        evs.append((voice, key + (key % 12 == 9), bool(rng.random() < 0.65)))
    if b == 0:
        evs = [(v, 36 + (7 * v) % 49, True) for v in range(0, n, 2)]
    script.append(evs)
    sizes.append(int(rng.choice([256, 256, 256, 100, 37, 1])))
ob = O.Bank.welsh(params)
want = []
for evs, fr in zip(script, sizes):
    if evs:
        ob.note_events(T.note_events(evs))
    want.append(ob.render(fr))
want = np.concatenate(want, axis=1)
ctx = E.Context(0)
old = (ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves)
starts = np.cumsum([0] + sizes)
for form in ("tp", "any", "split", "per-kind"):
    ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
    ctx.split_max_waves = (1 << 20) if form == "split" else 0
    ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
    s = E.WelshSynth(ctx, params)
    blk = ctx.block(n, 256)
    got = []
    for evs, fr in zip(script, sizes):
        if evs:
            s.handle_midi_events(T.note_events(evs))
        s.generate_batch_values(blk, fr)
        got.append(blk.download(fr))
    got = np.concatenate(got, axis=1).astype(np.float64)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    v = int(np.argmax(rms))
    e = np.abs(got - want)[:, :, v].max(axis=0)
    per_block = [float(e[starts[b]:starts[b + 1]].max()) for b in range(blocks)]
    print(form, "worst voice", v, f"rms {rms.max():.2e}", "patch", v % 32, "per-block max |err|:", [f"{x:.0e}" for x in per_block])
    if form == "tp":
        for b, evs in enumerate(script):
            mine = [(vv, k, on) for vv, k, on in evs if vv == v or vv == T.ALL_VOICES]
            if mine:
                print("   block", b, "frames", sizes[b], "events for it:", mine)
    s.destroy(); blk.destroy()

# detail: the worst voice of the last form around the block where the error appears
v = int(sys.argv[2]) if len(sys.argv) > 2 else v
b0 = int(sys.argv[3]) if len(sys.argv) > 3 else 24
lo = int(starts[b0])
print("voice", v, "frames", lo - 3, "..", lo + 12)
for f in range(lo - 3, lo + 12):
    print(f, "got", [round(float(x), 6) for x in got[:, f, v]], "want", [round(float(x), 6) for x in want[:, f, v]], "ratio L", round(float(got[0, f, v] / want[0, f, v]), 6) if want[0, f, v] else None)
seg = slice(lo, lo + 2000)
num = float(np.dot(got[0, seg, v], want[0, seg, v]) / np.dot(want[0, seg, v], want[0, seg, v]))
print("least-squares gain got/want over the next 2000 frames:", num, " residual rms after gain:", float(np.sqrt(np.mean((got[0, seg, v] - num * want[0, seg, v]) ** 2))))
