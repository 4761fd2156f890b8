"""One seed of tests/test_emul_numerics.py::test_random_patches_emulated_device_arithmetic_against_the_oracle taken apart on the CPU: per-voice
error, the voices' levels, the worst voice's patch and its error by block.   python3 tools/random_emul_debug.py SEED [f32kind]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import patches as P, abi_types as T
from tests.emul import emul as E
from oracle import oracle as O
O.build(ref=False)
seed = int(sys.argv[1]); f32_kind = len(sys.argv) > 2 and sys.argv[2] == "1"
n, blocks, off_at = 32, 40, 24
lanes = np.arange(n, dtype=np.uint32)
rng = np.random.default_rng(seed)
patches = [P.random_welsh_patch(rng) for _ in range(8)]
params = (T.WelshParams * n)(*[patches[(i // 4) % 8] for i in range(n)])
keys = rng.integers(30, 96, size=n).astype(np.uint8)
keys[keys % 12 == 9] += 1
on, off = T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)
bo, be = O.Bank.welsh(params), E.Bank.welsh(params)
if f32_kind:
    be.set_f32_kind(True)
o, e = [], []
for b in range(blocks):
    if b == 0: bo.note_events(on); be.note_events(on)
    if b == off_at: bo.note_events(off); be.note_events(off)
    o.append(bo.render(256)); e.append(be.render(256))
o = np.concatenate(o, axis=1); e = np.concatenate(e, axis=1).astype(np.float64)
err = np.sqrt(np.mean((e - o) ** 2, axis=(0, 1))); lvl = np.sqrt(np.mean(o ** 2, axis=(0, 1))); peak = np.abs(o).max(axis=(0, 1))
for v in range(n):
    print("voice %2d patch %d key %2d  err %.2e  level %.3f  peak %.2f  err/max(1,level) %.2e" % (v, (v // 4) % 8, keys[v], err[v], lvl[v], peak[v], err[v] / max(1, lvl[v])))
v = int(np.argmax(err))
d = (e - o)[0, :, v].reshape(blocks, 256)
print("worst voice", v, "error by block:", " ".join("%.1e" % x for x in np.sqrt(np.mean(d ** 2, axis=1))))
print("its peak by block:", " ".join("%.2f" % x for x in np.abs(o[0, :, v].reshape(blocks, 256)).max(axis=1)))
def dump(st, pre=""):
    for name, _ in st._fields_:
        x = getattr(st, name)
        if isinstance(x, ctypes.Structure): dump(x, pre + name + ".")
        else: print("   ", pre + name, x)
dump(patches[(v // 4) % 8])
