"""One patch of the reference's library (tools/reference_patches_emul.py) taken apart: per-voice error and level, the worst voice's error by
block.   python3 tools/reference_patch_debug.py NAME [key ...]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T
from tests.emul import emul as E
from oracle import oracle as O
O.build(ref=False)
L = C.CDLL(os.path.join(REPO, "groove_amd", "host", "libgroove_host.so"))
L.gh_welsh_params_from_patch_json.argtypes = [C.c_char_p, C.POINTER(T.WelshParams), C.c_char_p, C.c_size_t]
p = T.WelshParams(); err = C.create_string_buffer(512)
assert L.gh_welsh_params_from_patch_json(open("/root/reference/assets/patches/welsh/%s.json" % sys.argv[1]).read().encode(), C.byref(p), err, 512) == 0
keys = np.array([int(k) for k in sys.argv[2:]] or [31, 50, 64, 86], dtype=np.uint8)
n = len(keys)
lanes = np.arange(n, dtype=np.uint32)
on, off = T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)
params = (T.WelshParams * n)(*([p] * n))
SR = int(os.environ.get("SR", "44100"))
bo, be = O.Bank.welsh(params, sr=SR), E.Bank.welsh(params, SR)
o, e = [], []
blocks = 60
for b in range(blocks):
    if b == 0: bo.note_events(on); be.note_events(on)
    if b == 40: bo.note_events(off); be.note_events(off)
    o.append(bo.render(256)); e.append(be.render(256))
o = np.concatenate(o, axis=1); e = np.concatenate(e, axis=1).astype(np.float64)
er = np.sqrt(np.mean((e - o) ** 2, axis=(0, 1))); lv = np.sqrt(np.mean(o ** 2, axis=(0, 1)))
for v in range(n):
    print("key %3d  err %.2e  level %.3f peak %.2f" % (keys[v], er[v], lv[v], np.abs(o[:, :, v]).max()))
v = int(np.argmax(er))
d = (e - o)[0, :, v]
print("worst key", keys[v], "by block:", " ".join("%.0e" % x for x in np.sqrt(np.mean(d.reshape(blocks, 256) ** 2, axis=1))))
i = int(np.argmax(np.abs(d) > 1e-6)); print("first frame with |err| > 1e-6:", i, "oracle", o[0, i - 2:i + 4, v], "emul", e[0, i - 2:i + 4, v])
