"""tests/test_gpu_deferred.py::test_allpass_stream_random_walks by hand for one seed: the calls it makes, and where the bus goes wrong.
    python3 tools/random_walk_debug.py <seed> [ap = 0]"""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P

seed, ap = int(sys.argv[1]), bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
probe = len(sys.argv) > 3
ctx = E.Context(0)
rng = np.random.default_rng(seed)
ctx.fx_allpass_stream = ap
n = 192
synths = [E.WelshSynth(ctx, P.welsh_voices(n, first_voice=7 * c)) for c in range(2)]
fxp = [(T.FxParams * n)(*[T.fx_params(cutoff_hz=600.0 + 400 * c + 9 * (i % 50), delay_seconds=0.02 + 0.015 * c, attenuation=0.8, reverb_seconds=0.5 + 0.4 * c) for i in range(n)]) for c in range(2)]
chains = [[E.Effect(ctx, T.FX_BIQUAD_LP12, fxp[0]), E.Effect(ctx, T.FX_DELAY, fxp[0]), E.Effect(ctx, T.FX_REVERB, fxp[0])],
          [E.Effect(ctx, T.FX_REVERB, fxp[1]), E.Effect(ctx, T.FX_DELAY, fxp[1])]]
rots = [[ctx.block(n, 256) for _ in range(3)] for _ in range(2)]
blocks = 26
frames_of = [int(rng.choice([256, 256, 256, 100, 37, 256, 1])) for _ in range(blocks)]
bus = ctx.bus(sum(frames_of))
for s in synths:
    s.handle_midi_events(P.note_on_all(n))
at = 0
for b, fr in enumerate(frames_of):
    first = True
    log = [f"block {b} frames {fr}:"]
    for c in range(2):
        blk = rots[c][b % 3]
        if rng.random() < 0.6:
            blk.wait_released(); log.append("wait_released")
        if rng.random() < 0.7:
            synths[c].generate_batch_values_async(blk, fr); log.append("render_async")
            if rng.random() < 0.6:
                blk.wait_ready(); log.append("wait_ready")
        else:
            synths[c].generate_batch_values(blk, fr); log.append("render")
        if probe:
            log.append(f"[dry max {float(np.abs(blk.download(fr)).max()):.3g}]")
        if rng.random() < 0.75:
            ctx.transform_chain(chains[c], blk, fr); log.append("chain")
        else:
            for e in chains[c]:
                e.transform_audio(blk, fr)
            log.append("stages")
        if probe:
            log.append(f"[wet max {float(np.abs(blk.download(fr)).max()):.3g}]")
        how = rng.random()
        if how < 0.55:
            ctx.mix_deferred(blk, fr, E._Slice(bus, at), accumulate=not first); log.append(f"mix_deferred acc={int(not first)}"); first = False
        elif how < 0.85:
            ctx.mix([blk], fr, E._Slice(bus, at), accumulate=not first); log.append(f"mix acc={int(not first)}"); first = False
        if rng.random() < 0.15:
            blk.download(fr); log.append("download block")
        if rng.random() < 0.7:
            blk.release(); log.append("release")
        log.append("|")
    if first:
        ctx.mix([], fr, E._Slice(bus, at)); log.append("silence")
    r = rng.random()
    if r < 0.08:
        bus.download(); log.append("download bus")
    elif r < 0.14:
        chains[0][2].control_set_param_by_index(T.CTL_FX_ATTENUATION, float(rng.random())); log.append("set attenuation")
    elif r < 0.18:
        for e in chains[int(rng.integers(2))]:
            e.reset()
        log.append("reset chain")
    got = bus.download()[at:at + fr] if probe else None
    print(" ".join(log), (f" bus max {float(np.abs(got).max()):.3g}" if probe else ""))
    at += fr
out = bus.download()
print("bus max", float(np.abs(out).max()), "per block:", [round(float(np.abs(out[sum(frames_of[:b]):sum(frames_of[:b + 1])]).max()), 3) for b in range(blocks)])
