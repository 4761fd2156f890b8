"""One seed of tests/test_gpu_random_inputs.py::test_random_patches_at_other_sample_rates on the emulated device arithmetic (CPU): per-voice
error and level, the worst voice's error by block, its patch.   python3 tools/random_sr_emul_debug.py SEED"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import patches as P, abi_types as T
from tests.emul import emul as E
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
on, off = T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)
bo, be = O.Bank.welsh(params, sr=sr), E.Bank.welsh(params, sr)
o, e = [], []
for b in range(blocks):
    if b == 0: bo.note_events(on); be.note_events(on)
    if b == off_at: bo.note_events(off); be.note_events(off)
    o.append(bo.render(256)); e.append(be.render(256))
o = np.concatenate(o, axis=1); e = np.concatenate(e, axis=1).astype(np.float64)
err = np.sqrt(np.mean((e - o) ** 2, axis=(0, 1))); lvl = np.sqrt(np.mean(o ** 2, axis=(0, 1)))
print("seed", seed, "sr", sr)
v = int(np.argmax(err / np.maximum(1, lvl)))
for u in range((v // 8) * 8, (v // 8) * 8 + 8):
    print("voice %2d key %2d err %.2e level %.3f peak %.2f  relative to max(1, level) %.2e" % (u, keys[u], err[u], lvl[u], np.abs(o[:, :, u]).max(), err[u] / max(1, lvl[u])))
d = (e - o)[0, :, v].reshape(blocks, 256)
print("worst voice", v, "error by block:", " ".join("%.0e" % x for x in np.sqrt(np.mean(d ** 2, axis=1))))
print("peak by block:", " ".join("%.2f" % x for x in np.abs(o[0, :, v].reshape(blocks, 256)).max(axis=1)))
p = patches[(v // 8) % 8]
for name in ("lfo_waveform", "lfo_routing", "lfo_frequency", "lfo_depth", "filter_cutoff_hz", "filter_passband_ripple", "filter_cutoff_start", "filter_cutoff_end", "oscillator_2_sync"):
    print("   ", name, getattr(p, name))
print("    waves", p.oscillator_1.waveform, p.oscillator_2.waveform, "filter env", p.filter_envelope.attack, p.filter_envelope.decay, p.filter_envelope.sustain, p.filter_envelope.release)
