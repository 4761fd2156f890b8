"""tests/test_gpu_random_inputs.py::test_random_controls_on_a_sounding_bank_in_every_kernel_form by hand for one seed: per form the worst voice,
the block where its error appears and the control changes it received.   python3 tools/random_controls_debug.py <seed>"""
import os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

seed = int(sys.argv[1])
n, blocks = 96, 30
params, idx_of = P.welsh_voices_grouped(n)
lanes = np.arange(n, dtype=np.uint32)
keys = (36 + (7 * np.arange(n)) % 49).astype(np.uint8)
rng = np.random.default_rng(seed)
script, sizes = [], []
for b in range(blocks):
    changes = []
    for _ in range(int(rng.integers(0, 4)) if b else 0):
        idx = int(rng.choice([T.CTL_WELSH_DCA_GAIN, T.CTL_WELSH_DCA_PAN, T.CTL_WELSH_CUTOFF]))
        changes.append((idx, float(rng.uniform(0.1, 1.0)), T.ALL_VOICES if rng.random() < 0.25 else int(rng.integers(n))))
    script.append(changes)
    sizes.append(int(rng.choice([256, 256, 256, 100, 37, 1])))


def play(bank_events, bank_control, render):
    out = []
    for b in range(blocks):
        if b == 0 or b == 22:
            bank_events(T.note_events_np(lanes, keys, True))
        if b == 14:
            bank_events(T.note_events_np(lanes, keys, False))
        for idx, v, voice in script[b]:
            bank_control(idx, v, voice)
        out.append(render(sizes[b]))
    return np.concatenate(out, axis=1)


ob = O.Bank.welsh(params)
want = play(ob.note_events, lambda i, v, voice: ob.set_param(i, v, voice), ob.render)
ctx = E.Context(0)
old = (ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves)
starts = np.cumsum([0] + sizes)
for form in ("tp", "any", "split", "per-kind"):
    ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
    ctx.split_max_waves = (1 << 20) if form == "split" else 0
    ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
    s = E.WelshSynth(ctx, params)
    blk = ctx.block(n, 256)

    def render(fr):
        s.generate_batch_values(blk, fr)
        return blk.download(fr)

    got = play(s.handle_midi_events, lambda i, v, voice: s.control_set_param_by_index(i, v, voice=voice), render).astype(np.float64)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    v = int(np.argmax(rms))
    e = np.abs(got - want)[:, :, v].max(axis=0)
    print(form, "worst voice", v, "patch", int(idx_of[v]) % 32, f"rms {rms.max():.2e}", "per-block max |err|:", [f"{float(e[starts[b]:starts[b + 1]].max()):.0e}" for b in range(blocks)])
    if form == "tp":
        for b, ch in enumerate(script):
            mine = [(i, round(val, 3), vo) for i, val, vo in ch if vo in (v, T.ALL_VOICES)]
            if mine:
                print("   block", b, "frames", sizes[b], "controls:", mine)
    s.destroy(); blk.destroy()
