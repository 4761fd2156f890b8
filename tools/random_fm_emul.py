"""tests/test_gpu_instruments.py::test_random_fm_patches_against_the_oracle on the emulated device arithmetic (tests/emul), many seeds on the
CPU: the voices over their bar, with index, key and ratio.   python3 tools/random_fm_emul.py BASE N"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T
from tests.emul import emul as E
from oracle import oracle as O
O.build(ref=False)
base, count = int(sys.argv[1]), int(sys.argv[2])
n, blocks = 48, 40
lanes = np.arange(n, dtype=np.uint32)
worst, over = 0.0, []
for seed in range(base, base + count):
    rng = np.random.default_rng(seed)
    ps = []
    for _ in range(n):
        p = T.FmParams()
        p.ratio, p.depth, p.beta = float(rng.choice([0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 7.0, rng.uniform(0.3, 9.0)])), float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.05, 20.0))
        env = lambda: T.EnvelopeParams(0.0 if rng.random() < 0.2 else float(rng.uniform(0.001, 0.3)), float(rng.uniform(0.05, 1.5)),   # noqa: E731
                                       0.0 if rng.random() < 0.15 else float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.05, 1.0)))
        p.carrier_envelope, p.modulator_envelope = env(), env()
        if p.carrier_envelope.sustain == 0.0 and p.carrier_envelope.decay < 0.3:
            p.carrier_envelope.decay = 0.3
        p.dca_gain, p.dca_pan = float(rng.uniform(0.3, 1.0)), float(rng.uniform(-1.0, 1.0))
        ps.append(p)
    params = (T.FmParams * n)(*ps)
    keys = rng.integers(30, 96, size=n).astype(np.uint8)
    ob, be = O.Bank.fm(params), E.Bank.fm(params)
    got, want = [], []
    for b in range(blocks):
        if b in (0, 30):
            ev = T.note_events_np(lanes, keys, True); be.note_events(ev); ob.note_events(ev)
        if b == 20:
            ev = T.note_events_np(lanes, keys, False); be.note_events(ev); ob.note_events(ev)
        fr = int(rng.choice([256, 256, 256, 100, 7, 1]))
        got.append(be.render(fr)); want.append(ob.render(fr))
    got = np.concatenate(got, axis=1).astype(np.float64); want = np.concatenate(want, axis=1)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    bar = np.array([float(np.clip(2e-6 * p.beta * p.depth, 1e-5, 2e-5)) for p in ps])
    worst = max(worst, float((rms / bar).max()))
    for v in np.flatnonzero(rms > 0.5 * bar):
        p = ps[v]
        f = 440.0 * 2 ** ((int(keys[v]) - 69) / 12) * p.ratio
        over.append((seed, int(v), float(rms[v]), float(bar[v]), p.beta * p.depth, int(keys[v]), p.ratio, f / 44100.0))
print("seeds", base, "..", base + count - 1, "voices", count * n, "worst rms / bar %.3f" % worst)
for o in sorted(over, key=lambda x: -x[2] / x[3])[:25]:
    print("seed %d voice %2d rms %.2e bar %.1e index %5.2f key %d ratio %.4f modulator / SR %.4f" % o)
