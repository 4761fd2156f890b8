"""Delta-debugging of tests/test_gpu_random_inputs.py::test_random_note_event_sequences_in_every_kernel_form: the events ONE voice of a seed's
script received, replayed on a one-patch bank and shrunk greedily while the deviation from the oracle stays above the bar.
    python3 tools/event_minimise.py <seed> <voice>"""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

seed, voice = int(sys.argv[1]), int(sys.argv[2])
n, blocks = 96, 36
rng = np.random.default_rng(seed)
script, sizes = [], []
for b in range(blocks):
    evs = []
    for _ in range(int(rng.integers(0, 13)) if b else 0):
        v = T.ALL_VOICES if rng.random() < 0.06 else int(rng.integers(n)) if rng.random() < 0.7 else int(rng.integers(4))
        key = int(rng.integers(30, 96))
        evs.append((v, key + (key % 12 == 9), bool(rng.random() < 0.65)))
    if b == 0:
        evs = [(v, 36 + (7 * v) % 49, True) for v in range(0, n, 2)]
    script.append(evs)
    sizes.append(int(rng.choice([256, 256, 256, 100, 37, 1])))
mine = [[(0, k, on) for v, k, on in evs if v in (voice, T.ALL_VOICES)] for evs in script]
params = (T.WelshParams * 8)(*[P.welsh_patch(voice % 32)] * 8)
ctx = E.Context(0)


def deviation(ev_script, size_list):
    ob = O.Bank.welsh(params); s = E.WelshSynth(ctx, params); blk = ctx.block(8, 256)
    worst = 0.0
    for evs, fr in zip(ev_script, size_list):
        if evs:
            ob.note_events(T.note_events(evs)); s.handle_midi_events(T.note_events(evs))
        w = ob.render(fr); s.generate_batch_values(blk, fr); g = blk.download(fr).astype(np.float64)
        worst = max(worst, float(np.abs(g - w)[:, :, 0].max()))
    s.destroy(); blk.destroy()
    return worst


base = deviation(mine, sizes)
print("replayed on one voice: max |err|", base)
bar = 2e-5
cur, cur_sizes = [list(e) for e in mine], list(sizes)
changed = True
while changed and base > bar:
    changed = False
    for b in range(len(cur) - 1, -1, -1):          # drop whole blocks from the end, then single events
        trial, trial_sizes = cur[:b] + cur[b + 1:], cur_sizes[:b] + cur_sizes[b + 1:]
        if trial and deviation(trial, trial_sizes) > bar:
            cur, cur_sizes, changed = trial, trial_sizes, True
    for b in range(len(cur)):
        for i in range(len(cur[b]) - 1, -1, -1):
            trial = [list(e) for e in cur]
            del trial[b][i]
            if deviation(trial, cur_sizes) > bar:
                cur, changed = trial, True
print("minimal script (block: frames, events):")
for evs, fr in zip(cur, cur_sizes):
    print("  ", fr, evs)
print("its max |err|:", deviation(cur, cur_sizes))
