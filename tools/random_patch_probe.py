"""Random Welsh patches (continuous parameters drawn from a seed, not the 32 benchmark patches) through every kernel form against the f64
oracle: per-voice RMS error over a short timeline with a note-off.  Exploration tool behind tests/test_gpu_random_inputs.py's random test.
    python3 tools/random_patch_probe.py [seeds = 20] [voices per seed = 64]"""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, entities as E, patches as P
from oracle import oracle as O

random_patch = P.random_welsh_patch


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx = E.Context(0)
    old = (ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves)
    blocks, off_at = 40, 24
    worst, top = {}, []
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        patches = [random_patch(rng) for _ in range(8)]
        params = (T.WelshParams * n)(*[patches[(i // 8) % 8] for i in range(n)])     # runs of eight voices per patch
        keys = rng.integers(30, 96, size=n).astype(np.uint8)
        keys[keys % 12 == 9] += 1   # no A: 55 and 110 Hz are rational in SR 44,100 and a square's edge then lands EXACTLY on a frame (docs/DSP_SPEC.md section 2: ties)
        lanes = np.arange(n, dtype=np.uint32)
        ob = O.Bank.welsh(params)
        ob.note_events(T.note_events_np(lanes, keys, True))
        want = []
        for b in range(blocks):
            if b == off_at:
                ob.note_events(T.note_events_np(lanes, keys, False))
            want.append(ob.render(256))
        want = np.concatenate(want, axis=1)
        for form in ("tp", "any", "split", "per-kind"):
            ctx.time_parallel_max_voices = old[0] if form == "tp" else 0
            ctx.split_max_waves = (1 << 20) if form == "split" else 0
            ctx.pipeline_min_waves = 1 if form == "per-kind" else old[2]
            s = E.WelshSynth(ctx, params)
            blk = ctx.block(n, 256)
            s.handle_midi_events(T.note_events_np(lanes, keys, True))
            got = []
            for b in range(blocks):
                if b == off_at:
                    s.handle_midi_events(T.note_events_np(lanes, keys, False))
                s.generate_batch_values(blk, 256)
                got.append(blk.download(256))
            got = np.concatenate(got, axis=1).astype(np.float64)
            rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
            sig = np.sqrt(np.mean(want ** 2, axis=(0, 1)))
            j = int(np.argmax(rms))
            worst[(seed, form)] = (float(rms.max()), j, float(sig[j]))
            if form == "any":
                pj = patches[(j // 8) % 8]
                top.append((float(rms.max()), seed, j, int(keys[j]), f"w {pj.oscillator_1.waveform}/{pj.oscillator_2.waveform} sync {pj.oscillator_2_sync} lfo {pj.lfo_waveform}/{pj.lfo_routing} f {pj.lfo_frequency:.2f} d {pj.lfo_depth:.2f} "
                            f"cutoff {pj.filter_cutoff_hz:.0f} ripple {pj.filter_passband_ripple:.2f} sweep {pj.filter_cutoff_start:.2f}->{pj.filter_cutoff_end:.2f} signal {float(sig[j]):.2e}"))
            if rms.max() > 1e-5 or not np.isfinite(got).all():
                pj = patches[(j // 8) % 8]
                print(f"seed {seed} form {form}: voice {j} rms {rms.max():.3e} (signal {sig[j]:.3e}) key {keys[j]} w1 {pj.oscillator_1.waveform} w2 {pj.oscillator_2.waveform} sync {pj.oscillator_2_sync} "
                      f"lfo {pj.lfo_waveform}/{pj.lfo_routing} f {pj.lfo_frequency:.2f} d {pj.lfo_depth:.2f} cutoff {pj.filter_cutoff_hz:.0f} ripple {pj.filter_passband_ripple:.2f} start {pj.filter_cutoff_start:.2f} end {pj.filter_cutoff_end:.2f} "
                      f"amp {pj.amp_envelope.attack:.3f}/{pj.amp_envelope.decay:.2f}/{pj.amp_envelope.sustain:.2f}/{pj.amp_envelope.release:.2f}", flush=True)
            s.destroy(); blk.destroy()
    ctx.time_parallel_max_voices, ctx.split_max_waves, ctx.pipeline_min_waves = old
    for form in ("tp", "any", "split", "per-kind"):
        v = [worst[(s, form)][0] for s in range(seeds)]
        print(f"{form:9s} worst voice RMS over {seeds} seeds: max {max(v):.3e}  median {float(np.median(v)):.3e}")
    for t in sorted(top, reverse=True)[:6]:
        print("worst voices (all-kinds form): rms %.2e seed %d voice %d key %d  %s" % t)
    ctx.close()


main()
